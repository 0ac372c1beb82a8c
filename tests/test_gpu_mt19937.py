"""Device-side regeneration of the legacy NumPy MT19937 draws (MAUS_PERT_MT19937): the H matrices built
from a NumPy state on the device must be bit-identical to those built from the host's own
np.random.rand(N,N) draws (MAUS_PERT_UNIFORM), so the LU solutions are bit-identical too (GPU only)."""
import numpy as np
import pytest

import scenarios

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from adaptive_matrix_solver_amd import Context
    c = Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("n,count,pre_bytes,legacy", [(33, 5, 0, False), (100, 7, 4, False), (256, 4, 12, True),
                                                     (17, 3, 4, False), (8, 6, 0, False), (312, 3, 0, False), (640, 3, 4, False)])
def test_mt19937_device_draws_equal_host_draws(ctx, n, count, pre_bytes, legacy):
    from adaptive_matrix_solver_amd import _cabi
    A = scenarios.ginibre(n, n + 1, 1.0)
    ctx.set_matrix(A)
    ctx.pop_reserve(count)
    rng = np.random.default_rng(n)
    V = rng.standard_normal((count, n)) + 1j * rng.standard_normal((count, n))
    slots = list(range(count))
    shift = (rng.standard_normal(count) + 1j * rng.standard_normal(count)) * 0.2
    psi = np.full(count, 1e-3)                     # large enough that every draw changes H visibly
    np.random.seed(1000 + n)
    np.random.rand(777)                            # arbitrary stream position
    if pre_bytes:
        np.random.bytes(pre_bytes)                 # odd word position: doubles straddle block boundaries
    start = np.random.get_state()
    # host draws, exactly as the reference consumes them (a swallowed GMRES attempt draws one pair first)
    U = np.empty((count, 2, n, n))
    for k in range(count):
        if legacy:
            np.random.rand(n, n); np.random.rand(n, n)
        U[k, 0] = np.random.rand(n, n)
        U[k, 1] = np.random.rand(n, n)
    end_host = np.random.get_state()
    ctx.pop_put(0, slots, V)
    st1 = ctx.shifted_lu_solve(slots, shift, psi, pert_mode=_cabi.PERT_UNIFORM, pert_data=U)
    W1 = ctx.pop_get(2, slots, n)
    wpc = 4 * n * n * (2 if legacy else 1)
    lead = wpc - 4 * n * n
    ctx.pop_put(0, slots, V)
    st2 = ctx.shifted_lu_solve(slots, shift, psi, pert_mode=_cabi.PERT_MT19937,
                               pert_data=(start, wpc, lead, np.arange(count, dtype=np.int32)))
    W2 = ctx.pop_get(2, slots, n)
    assert np.all(st1 == 0) and np.all(st2 == 0)
    assert np.array_equal(W1, W2), "device-regenerated draws differ from np.random.rand"
    # and the host-side jump lands exactly where the host draws left the stream
    key, pos = _cabi.mt19937_jump(start[1], start[2], wpc * count)
    assert np.array_equal(key, end_host[1]) and pos == end_host[2]
    # perturbation really matters at this psi (guards against both paths silently dropping it)
    ctx.pop_put(0, slots, V)
    ctx.shifted_lu_solve(slots, shift, psi, pert_mode=_cabi.PERT_NONE)
    W0 = ctx.pop_get(2, slots, n)
    assert not np.allclose(W0, W1, rtol=1e-9, atol=0)


def test_sharded_ordinals_pick_the_right_substreams(ctx):
    """A rank that owns only some candidates of the run regenerates exactly their draws."""
    from adaptive_matrix_solver_amd import _cabi
    n, count = 64, 9
    A = scenarios.ginibre(n, 5, 1.0)
    ctx.set_matrix(A)
    ctx.pop_reserve(count)
    rng = np.random.default_rng(3)
    V = rng.standard_normal((count, n)) + 1j * rng.standard_normal((count, n))
    shift = np.zeros(count, dtype=np.complex128)
    psi = np.full(count, 1e-2)
    np.random.seed(5)
    start = np.random.get_state()
    slots = list(range(count))
    ctx.pop_put(0, slots, V)
    ctx.shifted_lu_solve(slots, shift, psi, pert_mode=_cabi.PERT_MT19937,
                         pert_data=(start, 4 * n * n, 0, np.arange(count, dtype=np.int32)))
    Wall = ctx.pop_get(2, slots, n)
    mine = [2, 3, 7]
    ctx.pop_put(0, slots, V)
    ctx.shifted_lu_solve(mine, shift[mine], psi[mine], pert_mode=_cabi.PERT_MT19937,
                         pert_data=(start, 4 * n * n, 0, np.array(mine, dtype=np.int32)))
    Wsub = ctx.pop_get(2, mine, n)
    assert np.array_equal(Wsub, Wall[mine])


@pytest.mark.parametrize("name,iters", [("eig64", 8), ("eig48u", 8), ("lin32f", 6)])
def test_trajectory_with_device_draws(name, iters):
    """Whole loop bodies with pert_mode='mt19937' against the oracle: same bar as the uploaded-draws mode."""
    from test_gpu_step_parity import compare, oracle_run, product_run
    ref, anorm = oracle_run(name, iters)
    got = product_run(name, iters, pert_mode="mt19937", gmres_compat="scipy-legacy")
    compare(ref, got, anorm, name + "-mt19937")


@pytest.mark.parametrize("n,count", [(96, 200), (64, 600)])
def test_two_sub_batch_streams_keep_their_own_generator_state(ctx, n, count, monkeypatch):
    """MAUS_LU_STREAMS=2: >= 128 solves run as two sub-batches on two streams: each needs its own generator buffers (the host prepares
    the second while the first still reads its states).  Same bits as the host-drawn path, with a psi large enough
    for the perturbation to reach the leading digits."""
    from adaptive_matrix_solver_amd._cabi import PERT_MT19937, PERT_UNIFORM
    monkeypatch.setenv("MAUS_LU_STREAMS", "2")
    rng = np.random.default_rng(11)             # (64, 600): more solves than the 512-matrix workspace -> two chunks
    A = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) / np.sqrt(n)
    V = rng.standard_normal((count, n)) + 1j * rng.standard_normal((count, n))
    ctx.set_matrix(A)
    ctx.pop_reserve(count)
    slots = list(range(count))
    ctx.pop_put(0, slots, V)
    lam = (rng.standard_normal(count) + 1j * rng.standard_normal(count)) * 0.2
    psi = np.full(count, 0.05)
    np.random.seed(77)
    np.random.rand(5)
    st = np.random.get_state()
    U = np.empty((count, 2, n, n))
    for k in range(count):
        U[k, 0] = np.random.rand(n, n)
        U[k, 1] = np.random.rand(n, n)
    for rep in range(3):            # the hazard is a race: give it a few chances
        s1 = ctx.shifted_lu_solve(slots, lam, psi, 0, PERT_UNIFORM, U)
        W1 = ctx.pop_get(2, slots, n)
        s2 = ctx.shifted_lu_solve(slots, lam, psi, 0, PERT_MT19937, (st, 4 * n * n, 0, np.arange(count, dtype=np.int32)))
        W2 = ctx.pop_get(2, slots, n)
        assert (s1 == 0).all() and (s2 == 0).all()
        assert np.array_equal(W1, W2), f"rep {rep}: {np.argwhere(np.any(W1 != W2, axis=1)).ravel()[:8]}"


@pytest.mark.parametrize("pert_mode", ["uniform", "mt19937"])
def test_e4_tiny_norm_reinit_moves_the_draws_of_the_candidates_behind_it(pert_mode):
    """E4 (AMS:283): a relaxed update with a tiny norm re-initialises the vector from 2 x rand(N), which shifts the
    rand(N,N) draws of every candidate stepped after it.  Forced here with a matrix scaled by 1e13 and alpha = 1 for two
    of eight candidates (their update is w itself, ||w|| ~ 1e-12).  With device-regenerated draws the run must be
    restarted behind the event exactly as with host draws: bookkeeping, both streams and the vectors against the
    oracle.  psi is escalated so that the perturbation reaches the leading digits of H (a candidate computed from the
    wrong stream position would differ visibly)."""
    import random
    import snapshot
    import scenarios
    from oracle import maus_oracle as orc
    from adaptive_matrix_solver_amd.engine import DeviceEngine
    from adaptive_matrix_solver_amd.solver import ProblemType, SolutionCandidate
    n, P = 40, 8
    A = scenarios.ginibre(n, 40, 1.0) * 1e13
    strat = {"overall_psi_aggression_factor": 1e31, "max_psi_retries": 25, "current_convergence_threshold": 1e-8,
             "convergence_tolerance": 1e-8}                      # psi = 1e11: 1 % of the entries
    hot = (2, 5)
    orc.seed_all(77)
    oc = [orc.new_candidate(A, orc.EIGENVALUE, n) for _ in range(P)]
    know_o = {"local_solver_preference": orc.DIRECT, "is_sparse_problem": False, "is_hermitian": False}
    ref = []
    for it in range(2):
        for k, c in enumerate(oc):
            if it == 0 and k in hot:
                c.alpha = np.complex128(1.0)
            orc.candidate_step(c, A, None, strat, know_o)
        ref.append([(c.state, c.stuck, c.retries, c.resets, complex(c.lam), c.v.copy(), c.resid) for c in oc] + [snapshot.rng_digest()])
    np.random.seed(77); random.seed(77); SolutionCandidate._candidate_id_counter = 0
    eng = DeviceEngine(pert_mode=pert_mode)
    pc = [SolutionCandidate(A, ProblemType.EIGENVALUE, n, engine=eng) for _ in range(P)]
    know = {"local_solver_preference": "direct_solve", "is_sparse_problem": False, "is_hermitian": False}
    for it in range(2):
        if it == 0:
            for k in hot:
                pc[k].alpha_local_step = np.complex128(1.0)
        eng.step(pc, A, None, strat, know)
        for k, c in enumerate(pc):
            r = ref[it][k]
            assert (c.state.value, c.stuck_counter, c.local_psi_retries_needed, c.num_resets) == r[:4], (it, k)
            assert abs(complex(c.lambda_k) - r[4]) <= 1e-9 * abs(r[4]), (it, k)
            v = np.asarray(c.v_k)
            assert np.linalg.norm(v - r[5]) <= 1e-7 * np.linalg.norm(r[5]), (it, k, np.linalg.norm(v - r[5]))
            assert abs(c.residual_k - r[6]) <= 1e-6 * r[6], (it, k)
        assert snapshot.rng_digest() == ref[it][P], it
    # the event really happened: the hot candidates hold un-normalised re-initialised vectors after the first step
    assert all(abs(np.linalg.norm(ref[0][k][5]) - 1.0) > 1e-3 for k in hot)
