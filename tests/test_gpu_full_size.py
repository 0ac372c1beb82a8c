"""Parity at BASELINE.json's full problem sizes (GPU box only).

Where the CPU oracle finishes in seconds it is run side by side (n=1024 eig loop bodies); at n=2048 (SVD),
n=4096 and n=8192 the HIP path is checked through size-independent properties computed with
NumPy on the host: the shifted solve round-trips (H w = v), Rayleigh quotients and residual norms
agree with a host GEMM, GMRES iterates satisfy SciPy's own stopping rule, the Hermitian match picks
the planted eigenvector, and the NumPy stream ends where really drawing the 4N^2 words per attempt
would leave it.  Candidate counts are the per-GPU shares of the BASELINE configs, trimmed where only
the host-side check (not the device) would otherwise dominate the run time.
"""
import random

import numpy as np
import pytest

import scenarios
from oracle import maus_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, scope="module")
def _blas_threads():
    """The host-side checks are BLAS-heavy; a GPU box hands one GPU's share of cores (16) to the job, and 64+ spinning
    BLAS threads on 16 cores are several times slower than 16."""
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        yield
        return
    with threadpool_limits(limits=16):
        yield


@pytest.fixture(scope="module")
def ctx():
    from adaptive_matrix_solver_amd import Context
    c = Context(0)
    yield c
    c.close()


# ---------------------------------------------------------------------------------------------
# configs[1]: 1024 x 1024 dense non-Hermitian eig -- whole loop bodies against the oracle
# (device regenerates the reference's MT19937 draws; the oracle really draws them)
# ---------------------------------------------------------------------------------------------
def test_eig1024_steps_against_oracle():
    import test_gpu_step_parity as sp
    scenarios.TRAJECTORIES["eig1024"] = dict(kind="eig", build=("ginibre", 1024, 1024, None), P=32, iters=2, seed=4321, tol=1e-8)
    try:
        ref, anorm = sp.oracle_run("eig1024", 2)
        got = sp.product_run("eig1024", 2)
        sp.compare(ref, got, anorm, "eig1024")
    finally:
        scenarios.TRAJECTORIES.pop("eig1024", None)


# ---------------------------------------------------------------------------------------------
# configs[3] (Hermitian shortcut) as whole loop bodies at a size where the oracle's eigh-per-candidate still finishes in
# seconds; the 8192 x 8192 run through MAUS_Solver is tools/c4_run.py (profiles/r02_c4_hermitian_8192_end_to_end.txt)
# ---------------------------------------------------------------------------------------------
def test_herm384_steps_against_oracle():
    import test_gpu_step_parity as sp
    scenarios.TRAJECTORIES["herm384"] = dict(kind="eig", build=("hermitian", 384, 384), P=24, iters=3, seed=9, tol=1e-8)
    try:
        ref, anorm = sp.oracle_run("herm384", 3)
        got = sp.product_run("herm384", 3)
        sp.compare(ref, got, anorm, "herm384", tie_tol=1e-13)
    finally:
        scenarios.TRAJECTORIES.pop("herm384", None)


# ---------------------------------------------------------------------------------------------
# configs[4]: 2048 x 2048 SVD, cond ~ 1e8 -- the alternating power step (AMS:233-242) and the SVD residual
# (AMS:299-301) of one GPU's share of the 512 candidates against batched NumPy.  (Whole SVD loop bodies are
# compared with the oracle at 64 x 48 in test_gpu_step_parity.py; at 2048^2 the oracle's per-candidate matvecs
# take minutes on the box's host share.)
# ---------------------------------------------------------------------------------------------
def test_svd2048_power_steps_against_numpy(ctx):
    from adaptive_matrix_solver_amd._cabi import KIND_SVD, POP_U, POP_X
    n, P = 2048, 64
    A = scenarios.prescribed_svd(n, n, 77, -8.0)
    rng = np.random.default_rng(5)
    V = rng.standard_normal((P, n)) + 1j * rng.standard_normal((P, n))
    V /= np.linalg.norm(V, axis=1)[:, None]
    slots = list(range(P))
    ctx.set_matrix(A)
    ctx.pop_reserve(P)
    ctx.pop_put(POP_X, slots, V)
    Vh = V.copy()
    for it in range(3):
        norms = ctx.svd_power_step(slots)
        T = Vh @ A.T                                              # rows: A v
        s1 = np.linalg.norm(T, axis=1)
        Uh = T / np.where(s1 > 1e-10, s1, 1.0)[:, None]
        S = Uh @ A.conj()                                         # rows: A^H u
        s2 = np.linalg.norm(S, axis=1)
        vin = np.linalg.norm(Vh, axis=1)
        Vh = S / np.where(s2 > 1e-10, s2, 1.0)[:, None]
        assert np.allclose(norms[:, 0], vin, rtol=1e-12)
        assert np.allclose(norms[:, 1], s1, rtol=1e-11)
        assert np.allclose(norms[:, 2], np.linalg.norm(Uh, axis=1), rtol=1e-12)
        assert np.allclose(norms[:, 3], s2, rtol=1e-11)
        Xd = ctx.pop_get(POP_X, slots, n)
        Ud = ctx.pop_get(POP_U, slots, n)
        assert np.linalg.norm(Xd - Vh) <= 1e-10 * np.linalg.norm(Vh)
        assert np.linalg.norm(Ud - Uh) <= 1e-10 * np.linalg.norm(Uh)
        # sigma estimates never exceed sigma_max = 1 and increase towards it
        assert (norms[:, 3] <= 1.0 + 1e-12).all()
    sig = norms[:, 3].astype(np.complex128)
    res, fin = ctx.residual(KIND_SVD, slots, sig)
    res_h = np.linalg.norm(Vh @ A.T - sig.real[:, None] * Uh, axis=1) + np.linalg.norm(Uh @ A.conj() - sig.real[:, None] * Vh, axis=1)
    assert fin.all()
    assert np.allclose(res, res_h, rtol=1e-8, atol=1e-13)


# ---------------------------------------------------------------------------------------------
# the metric's configuration: n = 4096 dense non-Hermitian eig, direct path
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def eig4096():
    n = 4096
    return n, scenarios.ginibre(n, 4096, None)


def test_eig4096_solve_roundtrip_rayleigh_residual(ctx, eig4096):
    """H_k w_k = v_k with H_k = A - lam_k I + psi_k I (+ the 0.15 psi perturbation, 1e-21, below the check),
    lam_k the device's Rayleigh quotient; all compared with host GEMMs."""
    from adaptive_matrix_solver_amd._cabi import KIND_EIG, PERT_MT19937
    n, A = eig4096
    P = 48
    rng = np.random.default_rng(7)
    V = (rng.standard_normal((P, n)) + 1j * rng.standard_normal((P, n))) / np.sqrt(n)
    slots = list(range(P))
    ctx.set_matrix(A)
    ctx.pop_reserve(P)
    ctx.pop_put(0, slots, V)
    num, den = ctx.matvec_rayleigh(slots)
    AV = V @ A.T                                               # row k = A v_k
    num_h = np.einsum("ki,ki->k", V.conj(), AV)
    den_h = np.einsum("ki,ki->k", V.conj(), V)
    assert np.allclose(num, num_h, rtol=1e-12, atol=1e-13)
    assert np.allclose(den, den_h, rtol=1e-13)
    lam = num / den
    psi = np.full(P, 1e-20)
    np.random.seed(99)
    st = np.random.get_state()
    status = ctx.shifted_lu_solve(slots, lam, psi, 0, PERT_MT19937, (st, 4 * n * n, 0, np.arange(P, dtype=np.int32)))
    assert (status == 0).all()
    W = ctx.pop_get(2, slots, n)
    HW = W @ A.T - lam[:, None] * W + psi[:, None] * W
    rel = np.linalg.norm(HW - V, axis=1) / np.linalg.norm(V, axis=1)
    # backward-stable LU: residual ~ eps * growth * ||H|| ||w|| / ||v||; inverse iteration makes ||w|| large
    bound = 1e-13 * np.linalg.norm(A, 1) * np.linalg.norm(W, axis=1) / np.linalg.norm(V, axis=1)
    assert (rel <= np.maximum(bound, 1e-12)).all(), (rel.max(), bound.min())
    # relaxed update + residual kernel against the host
    alpha = np.full(P, 0.7 + 0j)
    nrm = ctx.relax_normalise(slots, alpha, True)
    Xn = (1 - 0.7) * V + 0.7 * W
    assert np.allclose(nrm, np.linalg.norm(Xn, axis=1), rtol=1e-12)
    Xn /= np.linalg.norm(Xn, axis=1)[:, None]
    res, fin = ctx.residual(KIND_EIG, slots, lam)
    res_h = np.linalg.norm(Xn @ A.T - lam[:, None] * Xn, axis=1)
    assert fin.all()
    assert np.allclose(res, res_h, rtol=1e-9, atol=1e-13)


def test_eig4096_device_mt19937_equals_host_draws(ctx, eig4096):
    """The regenerated perturbation is the reference's: same solver output as with the draws made by NumPy on the
    host and uploaded (PERT_UNIFORM), bit for bit, at the metric's size."""
    from adaptive_matrix_solver_amd._cabi import PERT_MT19937, PERT_UNIFORM
    n, A = eig4096
    P = 3
    rng = np.random.default_rng(17)
    V = (rng.standard_normal((P, n)) + 1j * rng.standard_normal((P, n))) / np.sqrt(n)
    slots = list(range(P))
    ctx.set_matrix(A)
    ctx.pop_reserve(P)
    ctx.pop_put(0, slots, V)
    lam = (rng.standard_normal(P) + 1j * rng.standard_normal(P)) * 0.3
    psi = np.full(P, 1e-3)                     # large psi: the perturbation reaches the leading bits of H
    np.random.seed(2024)
    np.random.rand(333)                        # odd position inside a block
    st = np.random.get_state()
    U = np.empty((P, 2, n, n))
    for k in range(P):
        U[k, 0] = np.random.rand(n, n)
        U[k, 1] = np.random.rand(n, n)
    end_state = np.random.get_state()
    s1 = ctx.shifted_lu_solve(slots, lam, psi, 0, PERT_UNIFORM, U)
    W1 = ctx.pop_get(2, slots, n)
    s2 = ctx.shifted_lu_solve(slots, lam, psi, 0, PERT_MT19937, (st, 4 * n * n, 0, np.arange(P, dtype=np.int32)))
    W2 = ctx.pop_get(2, slots, n)
    assert (s1 == 0).all() and (s2 == 0).all()
    assert np.array_equal(W1, W2)
    # and the host-side jump lands where drawing the words left the stream
    from adaptive_matrix_solver_amd._cabi import mt19937_jump
    key, pos = mt19937_jump(st[1], st[2], 4 * n * n * P)
    assert pos == end_state[2] and np.array_equal(key, end_state[1])


def test_eig4096_loop_body_bookkeeping_and_stream():
    """One loop body of the product at n=4096: integer bookkeeping invariants and the NumPy stream position
    (exactly 4N^2 words per dense attempt, AMS:49) -- checked by really drawing them."""
    from adaptive_matrix_solver_amd.solver import MAUS_Solver, ProblemType, SolutionCandidate
    n, P = 4096, 12
    A = scenarios.ginibre(n, 4096, None)
    np.random.seed(5)
    random.seed(5)
    SolutionCandidate._candidate_id_counter = 0
    solver = MAUS_Solver(A, ProblemType.EIGENVALUE, initial_num_candidates=P, global_convergence_tol=1e-8, quiet=True)
    st0 = np.random.get_state()
    py0 = random.getstate()
    solver._update_global_diagnostics(1)
    solver._adjust_global_strategy(1)
    solver.step_population()
    st1 = np.random.get_state()
    assert random.getstate() == py0                       # the step itself draws nothing from `random`
    cands = list(solver.candidates)
    assert [c.id for c in cands] == list(range(P))
    assert all(c.local_psi_retries_needed == 0 and c.num_resets == 0 for c in cands)
    np.random.set_state(st0)
    for _ in range(P):
        np.random.rand(n, n)
        np.random.rand(n, n)
    ref = np.random.get_state()
    assert st1[2] == ref[2] and np.array_equal(st1[1], ref[1])
    # eigen-residual reported == recomputed on the host
    for c in cands[:4]:
        v = np.asarray(c.v_k)
        r = np.linalg.norm(A @ v - c.lambda_k * v)
        assert abs(r - c.residual_k) <= 1e-9 * max(r, 1e-12)
        assert abs(np.linalg.norm(v) - 1.0) <= 1e-12


def test_lu_beyond_4096_same_pivots_as_lapack(ctx):
    """n = 5000: panels taller than 4096 rows take the 16-rows-per-thread / 2-column panel variant."""
    import scipy.linalg as sla
    n = 5000
    rng = np.random.default_rng(50)
    A = (rng.standard_normal((1, n, n)) + 1j * rng.standard_normal((1, n, n))) / np.sqrt(n)
    b = rng.standard_normal((1, n)) + 1j * rng.standard_normal((1, n))
    x, status, ipiv = ctx.lu_solve(A, b, want_ipiv=True)
    assert status[0] == 0
    lu, piv = sla.lu_factor(A[0])
    assert np.array_equal(ipiv[0], piv)
    ref = sla.lu_solve((lu, piv), b[0])
    # backward error: no worse than a few times LAPACK's own on the same factorisation order
    r_dev = np.linalg.norm(A[0] @ x[0] - b[0])
    r_ref = np.linalg.norm(A[0] @ ref - b[0])
    floor = np.finfo(float).eps * np.linalg.norm(A[0], 1) * np.linalg.norm(ref)
    assert r_dev <= 10.0 * max(r_ref, floor), (r_dev, r_ref, floor)
    assert np.linalg.norm(x[0] - ref) <= 1e-7 * np.linalg.norm(ref)


# ---------------------------------------------------------------------------------------------
# configs[2]: 4096 x 4096 linear system, GMRES + Jacobi
# ---------------------------------------------------------------------------------------------
def test_lin4096_gmres_jacobi(ctx):
    n, P = 4096, 128                                # 512 candidates = 4 sweeps of this size through the same kernels
    A, b = scenarios.wide_diag_system(n, 11, decades=3.0, offdiag=0.02)
    ctx.set_matrix(A)
    ctx.set_rhs(b)
    ctx.pop_reserve(P)
    slots = list(range(P))
    ctx.pop_put(0, slots, np.tile(b, (P, 1)))
    psi = np.full(P, 1e-19) * (10.0 ** (np.arange(P) % 3))
    ok = ctx.jacobi_check(np.zeros(P, dtype=np.complex128), psi)
    assert ok.all()
    info, inner, status = ctx.gmres(slots, np.zeros(P, dtype=np.complex128), psi, 1, np.ones(P, dtype=np.int32))
    assert (status == 0).all()
    X = ctx.pop_get(2, slots, n)
    # one system on the host with the oracle's restatement of SciPy's GMRES: same iteration count, same iterate
    H0 = A + psi[0] * np.eye(n)
    inv_d = 1.0 / np.diag(H0)
    xr, info_r, inner_r, _ = orc.gmres_restated(H0, b, b, inv_d)
    assert info[0] == info_r and inner[0] == inner_r
    assert np.linalg.norm(X[0] - xr) <= 1e-9 * np.linalg.norm(xr)
    # every converged candidate satisfies SciPy's stopping rule on the true residual
    R = b[None, :] - (X @ A.T + psi[:, None] * X)
    rel = np.linalg.norm(R, axis=1) / np.linalg.norm(b)
    assert (info == 0).all()
    assert (rel <= 1e-8 * (1 + 1e-6)).all(), rel.max()


# ---------------------------------------------------------------------------------------------
# configs[3]: 8192 x 8192 Hermitian -- the per-candidate part of the eigh shortcut (AMS:165-175) at
# one GPU's share (128 of 1024 candidates); the eigenvectors are planted so that no 8192^3 eigh runs
# on the host inside a test (the product calls scipy's eigh once per matrix)
# ---------------------------------------------------------------------------------------------
def test_herm8192_match_planted_eigenvectors(ctx):
    n, P = 8192, 128
    rng = np.random.default_rng(8)
    u = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    u /= np.linalg.norm(u)
    d = np.linspace(-3.0, 5.0, n)
    # A = Q D Q^H with the Householder reflector Q = I - 2 u u^H (Hermitian, unitary): O(n^2) to form
    Du = d * u
    A = np.diag(d).astype(np.complex128)
    A -= 2.0 * np.outer(u, Du.conj())
    A -= 2.0 * np.outer(Du, u.conj())
    A += 4.0 * np.vdot(u, Du).real * np.outer(u, u.conj())
    Q = np.eye(n, dtype=np.complex128) - 2.0 * np.outer(u, u.conj())
    ctx.set_matrix(A)
    ctx.set_eigvecs(Q)
    ctx.pop_reserve(P)
    slots = list(range(P))
    target = rng.choice(n, size=P, replace=False)
    X = Q[:, target].T.copy()
    X += 0.05 * (rng.standard_normal((P, n)) + 1j * rng.standard_normal((P, n))) / np.sqrt(n)
    X *= np.exp(1j * rng.uniform(0, 2 * np.pi, P))[:, None]
    ctx.pop_put(0, slots, X)
    idx, nrm = ctx.herm_match(slots)
    assert np.array_equal(idx, target)
    Xm = ctx.pop_get(0, slots, n)
    # the picked column replaces the candidate (normalised), and it is an eigenvector of A to rounding
    assert np.allclose(np.linalg.norm(Xm, axis=1), 1.0, atol=1e-12)
    from adaptive_matrix_solver_amd._cabi import KIND_EIG
    res, fin = ctx.residual(KIND_EIG, slots, d[target].astype(np.complex128))
    assert fin.all() and res.max() <= 1e-11 * np.abs(d).max() * np.sqrt(n)
    res_h = np.linalg.norm(Xm @ A.T - d[target][:, None] * Xm, axis=1)
    assert np.allclose(res, res_h, atol=1e-12)
