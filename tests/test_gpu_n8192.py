"""BASELINE.json configs[3] at its own size: the 8192 x 8192 Hermitian eigenproblem with the whole decomposition on the device
(AMS:155-181; csrc/herm.hip), one GPU's share of the candidates, and the LU path at the largest matrix it accepts.  Nothing here
runs an O(n^3) computation on the host: every check is O(n^2) per sampled column."""
import random

import numpy as np
import pytest
import scipy.linalg as sla

import scenarios

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps
N = 8192
P = 128                     # 1024 candidates sharded 8 ways


@pytest.fixture(scope="module")
def herm8192():
    """The solver of configs[3] as a user builds it (eigh_mode='auto' takes the device path at this size): the start-up
    diagnostics decompose the matrix on the device -- the condition number of a Hermitian matrix is max|lambda| / min|lambda| --
    and hand the decomposition to the engine for the shortcut of AMS:155-181."""
    from adaptive_matrix_solver_amd.solver import MAUS_Solver, ProblemType, SolutionCandidate
    A = scenarios.hermitian(N, N)
    np.random.seed(1234)
    random.seed(1234)
    SolutionCandidate._candidate_id_counter = 0
    s = MAUS_Solver(A, ProblemType.EIGENVALUE, initial_num_candidates=P, global_convergence_tol=1e-8, quiet=True)
    yield A, s
    s.engine.ctx.close()


def test_configs3_loop_body_converges_every_candidate(herm8192):
    """AMS:155-181: every candidate picks its column of V in its first step; bookkeeping as the reference leaves it (CONVERGED,
    stuck 0, retries 0, weight 1, alpha untouched, no draw from either stream inside the step)."""
    A, s = herm8192
    from adaptive_matrix_solver_amd.solver import SolutionCandidate
    st0, py0 = np.random.get_state(), random.getstate()
    s._update_global_diagnostics(1)
    s._adjust_global_strategy(1)
    s.step_population()
    st1 = np.random.get_state()
    assert st1[2] == st0[2] and np.array_equal(st1[1], st0[1]) and random.getstate() == py0
    cands = list(s.candidates)
    assert [c.id for c in cands] == list(range(P))
    assert all(c.state == SolutionCandidate.State.CONVERGED for c in cands)
    assert all(c.stuck_counter == 0 and c.local_psi_retries_needed == 0 and c.w_k == 1.0 for c in cands)
    assert all(complex(c.alpha_local_step) == 0.01 for c in cands)
    anorm = np.abs(A).sum(axis=0).max()
    for c in cands[:6]:
        v = np.asarray(c.v_k)
        assert abs(np.linalg.norm(v) - 1.0) <= 1e-12
        assert isinstance(c.lambda_k, (float, np.floating)) or np.imag(c.lambda_k) == 0.0          # eigh's eigenvalues are real
        r = np.linalg.norm(A @ v - c.lambda_k * v)
        assert r <= 200 * N * EPS * anorm and abs(r - c.residual_k) <= 20 * EPS * anorm       # both are rounding noise of an exact eigenpair
    assert max(c.residual_k for c in cands) <= 200 * N * EPS * anorm
    assert s.engine.tridiag_solver == "device"
    # the rest of the loop body: the distinct converged set (one entry per distinct eigenpair) and the spawn of AMS:528-549
    s._manage_candidates(1)
    s._update_global_diagnostics(2)
    lams = np.array([np.real(c.lambda_k) for c in cands])
    assert s.num_distinct_converged_solutions == len(np.unique(np.round(lams, 9)))
    assert len(s.candidates) >= s.num_distinct_converged_solutions


def test_device_eigendecomposition_at_8192(herm8192):
    """The decomposition the loop body above left on the device, with O(n^2) checks: residuals of sampled columns, orthogonality
    of a 256-column sample against the bound of a solver without reorthogonalisation (eps ||T|| / gap), LAPACK's phase convention,
    and the eigenvalues against dstebz on the same tridiagonal matrix."""
    A, s = herm8192
    eng = s.engine
    assert eng.tridiag_solver == "device" and eng._eig_cache is not None
    gap, resid, tnorm = eng.tridiag_diag
    w = np.asarray(eng._eig_cache[1])
    V = eng.ctx.get_eigvecs()
    assert V.shape == (N, N) and w.shape == (N,)
    assert np.all(np.diff(w) >= 0)
    assert np.abs(V[0].imag).max() == 0.0                               # Q e_1 = e_1: first row of V = first row of the real Z
    anorm = np.linalg.norm(A, "fro") / np.sqrt(N) * 2.0                 # ~ the spectral radius of a GUE-like matrix, O(n^2)
    k = np.unique(np.concatenate([np.linspace(0, N - 1, 24).astype(int), [0, 1, N // 2, N - 2, N - 1]]))
    R = A @ V[:, k] - V[:, k] * w[k][None, :]
    assert np.linalg.norm(R, axis=0).max() <= 100 * N * EPS * anorm
    # orthogonality: neighbours (where the bound is weakest) and a spread sample
    for idx in (np.arange(256), np.arange(N // 2 - 128, N // 2 + 128), np.linspace(0, N - 1, 256).astype(int)):
        Gm = V[:, idx].conj().T @ V[:, idx] - np.eye(len(idx))
        assert np.abs(Gm).max() <= 50 * EPS / gap, (np.abs(Gm).max(), EPS / gap)
    assert gap >= 1e-7 and resid <= 1e-13                               # the acceptance test of engine.device_eigh held
    # eigenvalues: dstebz on the device's own T for sampled index ranges (O(n) per bisection step each)
    d, e = eng.tridiag_de
    for lo in (0, 1000, N // 2, N - 8):
        ref = sla.eigvalsh_tridiagonal(d, e, select="i", select_range=(lo, lo + 7), lapack_driver="stebz")
        assert np.abs(ref - w[lo:lo + 8]).max() <= 8 * EPS * tnorm
    # T is unitarily similar to A: the two O(n^2) invariants
    assert abs(d.sum() - np.trace(A).real) <= 100 * N * EPS * anorm
    fro2 = (d * d).sum() + 2.0 * (e * e).sum()
    assert abs(fro2 - np.linalg.norm(A, "fro") ** 2) <= 1e-10 * fro2


def test_lu_solve_round_trip_at_8192():
    """n = 8192, the largest matrix of the direct path: every panel above 4096 rows runs the 16-rows-per-thread variant.  The
    first panel's pivots against a host factorisation of the first 16 columns (LAPACK's rule: max |re| + |im|, first index
    wins), the solution by its backward error."""
    from adaptive_matrix_solver_amd import Context
    rng = np.random.default_rng(8192)
    A = ((rng.standard_normal((1, N, N)) + 1j * rng.standard_normal((1, N, N))) / np.sqrt(N)).astype(np.complex128)
    b = rng.standard_normal((1, N)) + 1j * rng.standard_normal((1, N))
    ctx = Context(0)
    try:
        x, status, ipiv = ctx.lu_solve(A, b, want_ipiv=True)
    finally:
        ctx.close()
    assert status[0] == 0
    piv = ipiv[0]
    assert piv.shape == (N,) and np.all(piv >= np.arange(N)) and np.all(piv < N)
    Pn = A[0][:, :16].copy()
    for c in range(16):                                                  # unblocked zgetf2 on the first panel
        p = c + int(np.argmax(np.abs(Pn[c:, c].real) + np.abs(Pn[c:, c].imag)))
        assert piv[c] == p, (c, piv[c], p)
        Pn[[c, p]] = Pn[[p, c]]
        Pn[c + 1:, c] /= Pn[c, c]
        Pn[c + 1:, c + 1:] -= np.outer(Pn[c + 1:, c], Pn[c, c + 1:])
    r = np.linalg.norm(A[0] @ x[0] - b[0])
    bound = EPS * np.abs(A[0]).sum(axis=0).max() * np.linalg.norm(x[0]) * N
    assert r <= bound, (r, bound)
    assert np.isfinite(x[0]).all()
