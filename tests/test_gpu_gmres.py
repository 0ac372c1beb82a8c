"""Batched device GMRES(+Jacobi) against SciPy's gmres and the oracle's restatement of it
(GPU box only).  Same systems as the reference-captured fixtures (solve_cases.json)."""
import json
import os

import numpy as np
import pytest

import scenarios
from oracle import maus_oracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx():
    from adaptive_matrix_solver_amd import Context
    c = Context(0)
    yield c
    c.close()


def run_device(ctx, A, b, shift, psi, use_j, rhs_mode=1):
    ctx.set_matrix(A)
    ctx.set_rhs(b)
    k = len(shift)
    ctx.pop_reserve(k)
    slots = list(range(k))
    ctx.pop_put(0, slots, np.tile(b, (k, 1)))
    info, inner, status = ctx.gmres(slots, np.asarray(shift, dtype=np.complex128), np.asarray(psi, dtype=np.float64),
                                    rhs_mode, np.asarray(use_j, dtype=np.int32))
    X = ctx.pop_get(2, slots, A.shape[0])
    return info, inner, status, X


@pytest.mark.parametrize("key,stuck", [("gmres_n32_s0", 0), ("gmres_n32_s2", 2), ("gmres_n128_s0", 0), ("gmres_n128_s2", 2)])
def test_gmres_converging_cases(ctx, key, stuck):
    A, b = scenarios.solve_case_inputs(key)
    psi = 1e-19
    H = A + psi * np.eye(A.shape[0])
    inv_d = orc.jacobi_inverse_diagonal(H, stuck)
    xs, info_s = orc.gmres_scipy(H, b, b, inv_d)
    xr, info_r, inner_r, cyc_r = orc.gmres_restated(H, b, b, inv_d)
    info, inner, status, X = run_device(ctx, A, b, [0j], [psi], [1 if inv_d is not None else 0])
    assert info[0] == info_s == info_r and status[0] == 0
    assert inner[0] == inner_r, (inner[0], inner_r)
    if info_s != 0:
        # no Jacobi (stuck <= 1) on this spectrum: SciPy exhausts 50 x 20 iterations as well; the
        # reference then falls back to LU (fixture attempts == 0 via the fallback)
        assert inner[0] == 50 * 20
        return
    # stated tolerance: the iterate agrees with SciPy's to 1e-9 relative (both satisfy rtol=1e-8)
    assert np.linalg.norm(X[0] - xs) <= 1e-9 * np.linalg.norm(xs)
    assert np.linalg.norm(b - H @ X[0]) <= 1e-8 * np.linalg.norm(b) * (1 + 1e-6)
    # the reference-captured fixture itself (through the tol->rtol shim)
    gx = np.load(os.path.join(GOLD, "solve_cases.npz"))[key + "_x"]
    assert np.linalg.norm(X[0] - gx) <= 1e-9 * np.linalg.norm(gx)


def test_gmres_multi_cycle_convergence(ctx):
    """Unpreconditioned system that needs several restart cycles: exercises the ptol adaptation and
    the restart bookkeeping; inner-iteration and outcome must equal SciPy's path."""
    n = 80
    rng = np.random.default_rng(5)
    A = np.diag(np.linspace(1, 200, n)).astype(np.complex128) + 0.5 * scenarios.ginibre(n, 6, 1.0)   # 6 cycles / 111 inner
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    xs, info_s = orc.gmres_scipy(A, b, b, None)
    xr, info_r, inner_r, cyc_r = orc.gmres_restated(A, b, b, None)
    info, inner, status, X = run_device(ctx, A, b, [0j], [0.0], [0])
    assert cyc_r >= 2, "scenario should restart at least once"
    assert info[0] == info_s == info_r
    assert abs(int(inner[0]) - inner_r) <= 1, (inner[0], inner_r)
    if info_s == 0:
        assert np.linalg.norm(X[0] - xs) <= 1e-8 * np.linalg.norm(xs)


def test_gmres_exhausts_without_preconditioner(ctx):
    """cond ~1e7 diagonal spread, no Jacobi: 50 restart cycles x 20 inner iterations, info = 50
    (the reference then raises LinAlgError and falls back to LU)."""
    A, b = scenarios.solve_case_inputs("gmresfb_n32_s0")
    info, inner, status, X = run_device(ctx, A, b, [0j], [1e-19], [0])
    xs, info_s = orc.gmres_scipy(A + 1e-19 * np.eye(32), b, b, None)
    assert info_s == 50 and info[0] == 50
    assert inner[0] == 50 * 20
    # with Jacobi the same system converges in a handful of iterations
    info, inner, status, X = run_device(ctx, A, b, [0j], [1e-19], [1])
    assert info[0] == 0 and inner[0] < 40
    assert np.linalg.norm(b - A @ X[0]) <= 1e-8 * np.linalg.norm(b) * (1 + 1e-6)


def test_gmres_batched_mixed_candidates(ctx):
    """Several candidates with different shifts / psi / preconditioning advance through their own
    (cycle, column) states in one batched run and match per-candidate SciPy solves."""
    n = 96
    A, b = scenarios.wide_diag_system(n, 4321, decades=2.0)
    rng = np.random.default_rng(7)
    k = 9
    shift = (rng.standard_normal(k) + 1j * rng.standard_normal(k)) * 0.3
    psi = 10.0 ** rng.uniform(-19, -12, k)
    use_j = (np.arange(k) % 2).astype(np.int32)
    ctx.set_matrix(A)
    ctx.pop_reserve(k)
    V = rng.standard_normal((k, n)) + 1j * rng.standard_normal((k, n))
    ctx.pop_put(0, list(range(k)), V)
    info, inner, status = ctx.gmres(list(range(k)), shift, psi, 0, use_j)          # rhs = X[slot] (eig form)
    X = ctx.pop_get(2, list(range(k)), n)
    for i in range(k):
        H = (A - shift[i] * np.eye(n)) + psi[i] * np.eye(n)
        inv_d = (1.0 / np.diag(H)) if use_j[i] else None
        xs, info_s = orc.gmres_scipy(H, V[i], V[i], inv_d)
        xr, info_r, inner_r, _ = orc.gmres_restated(H, V[i], V[i], inv_d)
        assert info[i] == info_s, i
        if info_s == 0:
            assert inner[i] == inner_r, (i, inner[i], inner_r)
            assert np.linalg.norm(X[i] - xs) <= 1e-8 * np.linalg.norm(xs), i


def test_jacobi_gate(ctx):
    n = 16
    A = scenarios.ginibre(n, 3, 1.0)
    A[5, 5] = 0.0
    ctx.set_matrix(A)
    ok = ctx.jacobi_check(np.array([0j, 0j, A[2, 2]]), np.array([0.0, 1e-3, 0.0]))
    assert list(ok) == [False, True, False]          # zero diagonal / fixed by psi / shift hits a diagonal entry


def test_inverse_iterate_solver_gmres_vs_reference_fixtures():
    """The drop-in InverseIterateSolver.solve with preferred_method='iterative_gmres' against the
    fixtures captured from the reference through the tol->rtol shim: same attempts, same method
    trace, iterate within 1e-9, NumPy stream position identical."""
    import random
    import snapshot
    from adaptive_matrix_solver_amd.solver import GLOBAL_DEFAULT_PSI_EPSILON_BASE, InverseIterateSolver
    meta = json.load(open(os.path.join(GOLD, "solve_cases.json")))
    arrays = np.load(os.path.join(GOLD, "solve_cases.npz"))
    for case in meta["cases"]:
        if not case["key"].startswith(("gmres_n", "gmresfb", "gmresbig", "gmres_legacy", "direct_", "bigpsi")):
            continue
        tgt, rhs = scenarios.solve_case_inputs(case["key"])
        np.random.seed(case["seed"]); random.seed(case["seed"])
        compat = "rtol" if case["gmres_shim"] else "scipy-legacy"
        s = InverseIterateSolver(case["n"], GLOBAL_DEFAULT_PSI_EPSILON_BASE * case["aggr"], case["max_attempts"],
                                 case["pref"], False, gmres_compat=compat, pert_mode="uniform")
        x, att = s.solve(tgt, rhs, case["stuck"])
        assert att == case["attempts"], case["key"]
        assert snapshot.rng_digest() == case["rng"], case["key"]
        gx = arrays[case["key"] + "_x"]
        assert np.linalg.norm(x - gx) <= 1e-9 * np.linalg.norm(gx), case["key"]
        if case["key"] == "gmresfb_n32_s0":
            assert [t["method"] for t in s.last_trace] == ["iterative_gmres", "direct_solve"]
            assert s.last_trace[0]["info"] == 50 and s.last_trace[0]["inner"] == 1000
        if case["key"].startswith("gmresbig"):
            # escalated psi: GMRES ran against the materialised H_solve (random term included); dropping the term moves
            # the iterate by 3e-6 .. 9e-6 relative on these systems, far outside the 1e-9 above
            assert s.last_trace[0]["method"] == "iterative_gmres" and s.last_trace[0].get("dense") is True
            assert s.last_trace[0]["jacobi"] == (case["stuck"] > 1)
            if case["key"] == "gmresbig_n32_s0":
                assert [t["method"] for t in s.last_trace] == ["iterative_gmres", "direct_solve"]
                assert s.last_trace[0]["info"] == 50


def test_linear_fragile_trajectory_with_device_gmres():
    """lin32f ('Fragile' -> GMRES preferred) with the tol->rtol intent honoured on both sides:
    oracle with its GMRES restatement vs the device GMRES, bookkeeping and streams exact."""
    from test_gpu_step_parity import compare, oracle_run, product_run
    ref, anorm = oracle_run("lin32f", 8, gmres_mode="restated")
    got = product_run("lin32f", 8, pert_mode="uniform", gmres_compat="rtol")
    compare(ref, got, anorm, "lin32f-gmres")


def test_gmres_pert_dense_mode_against_oracle_and_device_draws(ctx):
    """maus_gmres_pert: GMRES against H_k = A - s_k I + psi_k I + 0.15 psi_k ((U1-.5)+i(U2-.5)) (AMS:49-52, 89) for
    several candidates at once (eig form: rhs = the candidate's vector, per-candidate shift, psi and Jacobi request):
    info / inner-iteration count / iterate against the oracle's GMRES restatement on the host-built H_k, and the
    device-regenerated draws ('mt19937') equal to the uploaded host draws ('uniform') bit for bit."""
    from adaptive_matrix_solver_amd._cabi import PERT_MT19937, PERT_UNIFORM
    n, k = 128, 7
    A, _ = scenarios.wide_diag_system(n, 909, decades=2.5)
    rng = np.random.default_rng(10)
    V = rng.standard_normal((k, n)) + 1j * rng.standard_normal((k, n))
    shift = (rng.standard_normal(k) + 1j * rng.standard_normal(k)) * 0.2
    psi = 10.0 ** rng.uniform(-6, -3, k)
    want = (np.arange(k) % 3 != 0).astype(np.int32)
    ctx.set_matrix(A)
    ctx.pop_reserve(k)
    slots = list(range(k))
    ctx.pop_put(0, slots, V)
    np.random.seed(31)
    np.random.rand(7)
    st = np.random.get_state()
    U = np.empty((k, 2, n, n))
    for i in range(k):
        U[i, 0] = np.random.rand(n, n)
        U[i, 1] = np.random.rand(n, n)
    info, inner, status, jac = ctx.gmres_pert(slots, shift, psi, 0, want, PERT_UNIFORM, U)
    X = ctx.pop_get(2, slots, n)
    info2, inner2, status2, jac2 = ctx.gmres_pert(slots, shift, psi, 0, want, PERT_MT19937,
                                                  (st, 4 * n * n, 0, np.arange(k, dtype=np.int32)))
    X2 = ctx.pop_get(2, slots, n)
    assert np.array_equal(info, info2) and np.array_equal(inner, inner2) and np.array_equal(jac, jac2)
    assert np.array_equal(X, X2)
    assert (status == 0).all() and np.array_equal(jac, want.astype(bool))
    for i in range(k):
        ps = np.complex128(psi[i])
        H = (A - shift[i] * np.eye(n)) + (ps * np.eye(n) + (U[i, 0] - 0.5 + 1j * (U[i, 1] - 0.5)) * ps * 0.15)
        inv_d = (1.0 / np.diag(H)) if want[i] else None
        xr, info_r, inner_r, _ = orc.gmres_restated(H, V[i], V[i], inv_d)
        assert info[i] == info_r, i
        if info_r == 0:
            assert inner[i] == inner_r, (i, inner[i], inner_r)
            assert np.linalg.norm(X[i] - xr) <= 1e-9 * np.linalg.norm(xr), i
            # and the shared-matrix mode (random term left out) is measurably different here
            H0 = (A - shift[i] * np.eye(n)) + ps * np.eye(n)
            x0, _, _, _ = orc.gmres_restated(H0, V[i], V[i], (1.0 / np.diag(H0)) if want[i] else None)
            assert np.linalg.norm(x0 - xr) > 1e-8 * np.linalg.norm(xr), i


@pytest.mark.parametrize("pert_mode", ["uniform", "mt19937"])
def test_candidate_steps_with_escalated_psi_take_the_dense_gmres(pert_mode):
    """update_solution_step with a strategy whose aggression factor puts psi at 1e-6 (reachable deep in the retry
    ladder, or by a caller's strat_params): GMRES preferred, Jacobi for the stuck candidates.  Against the oracle:
    bookkeeping, both RNG streams, iterates."""
    import random
    import snapshot
    from adaptive_matrix_solver_amd.engine import DeviceEngine
    from adaptive_matrix_solver_amd.solver import ProblemType, SolutionCandidate
    n, P = 48, 6
    A, b = scenarios.wide_diag_system(n, 777, decades=3.0)
    strat = {"overall_psi_aggression_factor": 1e14, "max_psi_retries": 25, "current_convergence_threshold": 1e-8,
             "convergence_tolerance": 1e-8}
    stucks = [2, 0, 3, 2, 0, 4]
    orc.seed_all(23)
    oc = [orc.new_candidate(A, orc.SOLVE_LINEAR_SYSTEM, n) for _ in range(P)]
    know_o = {"local_solver_preference": orc.GMRES, "is_sparse_problem": False, "is_hermitian": False}
    ref = []
    for it in range(3):
        for c, sk in zip(oc, stucks):
            if it == 0:
                c.stuck = sk
            orc.candidate_step(c, A, b, strat, know_o, gmres_mode="restated")
        ref.append([(c.state, c.stuck, c.retries, c.resets, c.x.copy(), c.resid) for c in oc] + [snapshot.rng_digest()])
    np.random.seed(23); random.seed(23); SolutionCandidate._candidate_id_counter = 0
    eng = DeviceEngine(pert_mode=pert_mode, gmres_compat="rtol")
    pc = [SolutionCandidate(A, ProblemType.SOLVE_LINEAR_SYSTEM, n, engine=eng) for _ in range(P)]
    know = {"local_solver_preference": "iterative_gmres", "is_sparse_problem": False, "is_hermitian": False}
    assert eng.pert_matters(np.array([1e-6]))[0] and not eng.pert_matters(np.array([1e-19]))[0]
    for it in range(3):
        if it == 0:
            for c, sk in zip(pc, stucks):
                c.stuck_counter = sk
        eng.step(pc, A, b, strat, know)
        for k, c in enumerate(pc):
            r = ref[it][k]
            assert (c.state.value, c.stuck_counter, c.local_psi_retries_needed, c.num_resets) == r[:4], (it, k)
            assert np.linalg.norm(np.asarray(c.x_k) - r[4]) <= 1e-8 * np.linalg.norm(r[4]), (it, k)
            assert abs(c.residual_k - r[5]) <= 1e-6 * max(r[5], 1e-12), (it, k)
        assert snapshot.rng_digest() == ref[it][P], it
