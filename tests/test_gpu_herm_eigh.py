"""Hermitian eigendecomposition with the reduction and the back-transformation on the device (AMS:161; csrc/herm.hip):
maus_herm_tridiag -> scipy.linalg.eigh_tridiagonal (LAPACK dstemr, host) -> maus_herm_backtransform, against
scipy.linalg.eigh (LAPACK zheevr), the reference's call."""
import numpy as np
import pytest
import scipy.linalg as sla

import scenarios

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps


@pytest.fixture()
def ctx():
    from adaptive_matrix_solver_amd import Context
    c = Context(0)
    yield c
    c.close()


def _device_eigh(ctx, A):
    ctx.set_matrix(A)
    d, e = ctx.herm_tridiag()
    w, Z = sla.eigh_tridiagonal(d, e)
    ctx.herm_backtransform(Z)
    return d, e, w, ctx.get_eigvecs()


@pytest.mark.parametrize("n", [1, 2, 3, 17, 64, 65, 100, 128, 200, 777])
def test_tridiagonalisation_and_eigenvectors_against_lapack(ctx, n):
    A = scenarios.hermitian(n, 1000 + n)
    anorm = max(np.linalg.norm(A, 2), 1e-300)
    d, e, w, V = _device_eigh(ctx, A)
    wl, Vl = sla.eigh(A)
    # (i) the tridiagonal matrix is unitarily similar to A: same spectrum as LAPACK's
    assert np.abs(w - wl).max() <= 40 * n * EPS * anorm
    # (ii) V diagonalises A and is unitary
    assert np.linalg.norm(A @ V - V * w[None, :]) <= 40 * n * EPS * anorm * np.sqrt(n)
    assert np.linalg.norm(V.conj().T @ V - np.eye(n)) <= 40 * n * EPS * np.sqrt(n)
    # (iii) LAPACK's phase convention: Q e_1 = e_1, so the first row of V is the first row of the real Z
    assert np.abs(V[0].imag).max() == 0.0
    # (iv) same eigenvectors as zheevr where the eigenvalue is well separated -- up to the sign of the column: dstemr fixes it
    # at the twist index of its factorisation, and that choice moves under a 1-ulp change of T (about one column in ten
    # flips between LAPACK's own T and the same T perturbed in the last bit: test_dstemr_signs_are_rounding below), so no
    # tridiagonalisation with another summation order can reproduce it
    gaps = np.full(n, np.inf)
    if n > 1:
        dw = np.diff(wl)
        gaps[:-1] = dw
        gaps[1:] = np.minimum(gaps[1:], dw)
    flips = 0
    for k in range(n):
        if gaps[k] < 1e-6 * anorm or abs(Vl[0, k]) < 1e-8:
            continue                                    # close pair / vanishing first component: the sign is rounding there
        tol = 200 * n * EPS * anorm / gaps[k]
        dplus, dminus = np.linalg.norm(V[:, k] - Vl[:, k]), np.linalg.norm(V[:, k] + Vl[:, k])
        assert min(dplus, dminus) <= tol, (k, dplus, dminus, tol)
        flips += dminus < dplus
    assert flips <= max(2, n // 3), f"{flips} of {n} columns with the opposite sign: more than rounding explains"


def test_only_the_lower_triangle_counts(ctx):
    """scipy.linalg.eigh(A) reads the lower triangle (lower=True): an input that is Hermitian only to np.allclose's tolerance
    (the reference's test, AMS:384) must give what LAPACK gives for its lower triangle."""
    n = 150
    A = scenarios.hermitian(n, 77)
    rng = np.random.default_rng(6)
    A = A + np.triu(1e-9 * (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))), 1)      # upper triangle disturbed
    A[np.diag_indices(n)] += 1e-9j
    assert np.allclose(A, A.conj().T)
    d, e, w, V = _device_eigh(ctx, A)
    wl = sla.eigvalsh(A, lower=True)
    assert np.abs(w - wl).max() <= 40 * n * EPS * np.linalg.norm(A, 2)
    L = np.tril(A, -1)
    Al = L + L.conj().T + np.diag(A.diagonal().real)
    assert np.linalg.norm(Al @ V - V * w[None, :]) <= 1e-12


def test_dstemr_signs_are_rounding():
    """Why (iv) compares modulo sign: LAPACK's own tridiagonal matrix against itself with d, e moved by one ulp."""
    from scipy.linalg import lapack
    n = 200
    A = scenarios.hermitian(n, 1200)
    _, d, e, _, info = lapack.zhetrd(A, lower=1)
    assert info == 0
    _, Z = sla.eigh_tridiagonal(d, e)
    rng = np.random.default_rng(1)
    _, Z2 = sla.eigh_tridiagonal(d * (1 + rng.choice([-1, 0, 1], n) * EPS), e * (1 + rng.choice([-1, 0, 1], n - 1) * EPS))
    flips = sum(np.linalg.norm(Z[:, k] + Z2[:, k]) < np.linalg.norm(Z[:, k] - Z2[:, k]) for k in range(n))
    assert flips > 0


def test_already_tridiagonal_and_diagonal_matrices(ctx):
    n = 50
    rng = np.random.default_rng(3)
    dd, ee = rng.standard_normal(n), rng.standard_normal(n - 1)
    T = np.diag(dd).astype(np.complex128) + np.diag(ee, -1) + np.diag(ee, 1)
    d, e, w, V = _device_eigh(ctx, T)
    assert np.allclose(d, dd, atol=1e-15) and np.allclose(np.abs(e), np.abs(ee), atol=1e-15)
    assert np.linalg.norm(T @ V - V * w[None, :]) <= 1e-12
    D = np.diag(dd).astype(np.complex128)
    d, e, w, V = _device_eigh(ctx, D)
    assert np.array_equal(d, dd) and not e.any()
    assert np.allclose(np.sort(dd), w)


def test_complex_subdiagonal_is_made_real(ctx):
    """A Hermitian tridiagonal matrix with complex off-diagonals: every reflector is a pure phase (x empty or zero, Im alpha != 0)."""
    n = 40
    rng = np.random.default_rng(4)
    dd = rng.standard_normal(n)
    ee = rng.standard_normal(n - 1) + 1j * rng.standard_normal(n - 1)
    T = np.diag(dd).astype(np.complex128) + np.diag(ee, -1) + np.diag(ee.conj(), 1)
    d, e, w, V = _device_eigh(ctx, T)
    assert np.allclose(np.abs(e), np.abs(ee), rtol=1e-14)
    assert np.abs(w - sla.eigvalsh(T)).max() <= 1e-13
    assert np.linalg.norm(T @ V - V * w[None, :]) <= 1e-12
    assert np.abs(V[0].imag).max() == 0.0


def test_hermitian_shortcut_through_the_device_decomposition():
    """Loop bodies of a Hermitian eigenproblem with eigh_mode='device' against the oracle (host zheevr): every candidate converges
    onto the same eigenpair, bookkeeping and both RNG streams exact."""
    import test_gpu_step_parity as sp
    scenarios.TRAJECTORIES["herm320dev"] = dict(kind="eig", build=("hermitian", 320, 320), P=24, iters=3, seed=9, tol=1e-8)
    try:
        ref, anorm = sp.oracle_run("herm320dev", 3)
        got = sp.product_run("herm320dev", 3, eigh_mode="device")
        sp.compare(ref, got, anorm, "herm320dev", tie_tol=1e-13)
    finally:
        scenarios.TRAJECTORIES.pop("herm320dev", None)


def test_reporting_prologue_through_the_device_reduction():
    """AMS:559 / 567 (SURVEY f-4): the "true solution" of a Hermitian eigenproblem and of an SVD problem from the tridiagonal matrix
    the device reduces to, against SciPy's eigvals / svd."""
    import random
    from adaptive_matrix_solver_amd.solver import MAUS_Solver, ProblemType, SolutionCandidate
    n = 260
    A = scenarios.hermitian(n, 5)
    np.random.seed(3); random.seed(3); SolutionCandidate._candidate_id_counter = 0
    s = MAUS_Solver(A, ProblemType.EIGENVALUE, initial_num_candidates=6, quiet=True, eigh_mode="device")
    vals = s._reference_solution()
    ref = sla.eigvals(A); ref.sort()
    assert vals.dtype == np.complex128 and np.abs(vals - ref).max() <= 1e-12
    s.engine.ctx.close()
    B = scenarios.prescribed_svd(200, 140, 9, -6.0)
    np.random.seed(3); random.seed(3); SolutionCandidate._candidate_id_counter = 0
    s = MAUS_Solver(B, ProblemType.SVD, initial_num_candidates=6, quiet=True, eigh_mode="device")
    sv = s._reference_solution()
    ref = sorted(sla.svd(B, compute_uv=False).tolist(), reverse=True)
    assert len(sv) == 140 and np.abs(np.array(sv) - np.array(ref)).max() <= 1e-13 * ref[0] * 200
    # ... and the engine's own matrix / population are untouched by the scratch context
    s.loop_body(1)
    s.engine.ctx.close()


@pytest.mark.parametrize("kind", ["identity", "zero", "projector", "rank1", "scaled_small", "scaled_large", "real_symmetric"])
def test_degenerate_and_scaled_inputs(ctx, kind):
    """Repeated eigenvalues (the eigenvectors are then any orthonormal basis of the eigenspaces: checked through the
    decomposition itself), a zero matrix, badly scaled entries, a real symmetric matrix."""
    n = 96
    rng = np.random.default_rng(11)
    Q = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))[0]
    if kind == "identity":
        A = np.eye(n, dtype=np.complex128) * 2.5
    elif kind == "zero":
        A = np.zeros((n, n), dtype=np.complex128)
    elif kind == "projector":
        A = Q[:, :30] @ Q[:, :30].conj().T
    elif kind == "rank1":
        A = 3.0 * np.outer(Q[:, 0], Q[:, 0].conj())
    elif kind == "scaled_small":
        A = scenarios.hermitian(n, 3) * 1e-120
    elif kind == "scaled_large":
        A = scenarios.hermitian(n, 3) * 1e+120
    else:
        B = rng.standard_normal((n, n))
        A = ((B + B.T) / 2).astype(np.complex128)
    A = (A + A.conj().T) / 2
    d, e, w, V = _device_eigh(ctx, A)
    scale = max(np.abs(A).max(), 1e-300)
    assert np.all(np.isfinite(w)) and np.all(np.isfinite(V))
    assert np.abs(w - sla.eigvalsh(A)).max() <= 1e-12 * scale * n
    assert np.linalg.norm(A @ V - V * w[None, :]) <= 1e-12 * scale * n
    assert np.linalg.norm(V.conj().T @ V - np.eye(n)) <= 1e-12 * n
    assert np.abs(V[0].imag).max() == 0.0


# ---- the tridiagonal eigenproblem itself on the device (maus_herm_tridiag_eig: bisection + twisted factorisation) --------------
@pytest.mark.parametrize("n", [2, 3, 64, 257, 1000, 2048])
def test_tridiagonal_eigenproblem_on_the_device(ctx, n):
    """Eigenvalues against LAPACK's to a few ulps of ||T||; the vectors through the back-transformation diagonalise A with
    residuals at rounding level and are orthogonal to eps ||T|| / gap; the function's own diagnostics say so."""
    from adaptive_matrix_solver_amd.engine import TRIDIAG_MAX_RESID, TRIDIAG_MIN_GAP
    A = scenarios.hermitian(n, 4000 + n)
    anorm = np.linalg.norm(A, 2)
    ctx.set_matrix(A)
    d, e = ctx.herm_tridiag()
    # LAPACK's own solvers differ from each other by 7-20 ulps of ||T|| on such matrices (dstebz, the same bisection, against
    # dstemr); the device values sit within half an ulp of dstebz's
    wl = sla.eigh_tridiagonal(d, e, eigvals_only=True, lapack_driver="stemr")
    wz = sla.eigh_tridiagonal(d, e, eigvals_only=True, lapack_driver="stebz")
    w, (gap, resid, tnorm) = ctx.herm_tridiag_eig(d, e)
    assert np.all(np.diff(w) >= 0)
    assert np.abs(w - wl).max() <= 64 * EPS * tnorm, np.abs(w - wl).max() / (EPS * tnorm)
    assert np.abs(w - wz).max() <= 4 * EPS * tnorm, np.abs(w - wz).max() / (EPS * tnorm)
    assert gap >= TRIDIAG_MIN_GAP and resid <= TRIDIAG_MAX_RESID, (gap, resid)
    assert abs(gap - np.diff(wl).min() / tnorm) <= 1e-3 * gap + 16 * EPS
    ctx.herm_backtransform(None)                                 # the Z the call left on the device
    V = ctx.get_eigvecs()
    assert np.abs(A @ V - V * w[None, :]).max() <= 60 * n * EPS * anorm
    orth = np.abs(V.conj().T @ V - np.eye(n)).max()
    assert orth <= max(50 * EPS / gap, 1e3 * EPS), (orth, gap)    # eps ||T|| / gap, the bound the acceptance rule relies on
    assert np.abs(V[0].imag).max() == 0.0                        # LAPACK's phase convention survives (Q e_1 = e_1)
    # the same columns as LAPACK's up to sign
    Vl = sla.eigh(A)[1]
    assert np.abs(np.abs(np.einsum("ij,ij->j", Vl.conj(), V)) - 1.0).max() <= 1e3 * EPS / gap


def test_clustered_spectrum_goes_to_the_host_solver(monkeypatch):
    """Repeated eigenvalues: the device vectors would not be orthogonal inside a cluster; device_eigh must notice (gap test) and
    take dstemr -- and the decomposition must be as good as ever."""
    from adaptive_matrix_solver_amd.engine import DeviceEngine
    n = 300
    rng = np.random.default_rng(5)
    Q = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))[0]
    lam = np.concatenate([np.full(40, 1.0), np.full(3, -2.0), np.linspace(2.0, 5.0, n - 43)])
    A = (Q * lam[None, :]) @ Q.conj().T
    A = (A + A.conj().T) / 2
    eng = DeviceEngine(0, eigh_mode="device")
    try:
        w = eng.device_eigh(A)
        assert eng.tridiag_solver == "host" and eng.tridiag_diag[0] < 1e-7
        V = eng.ctx.get_eigvecs()
        assert np.abs(np.sort(w) - np.sort(lam)).max() <= 1e-12
        assert np.abs(A @ V - V * w[None, :]).max() <= 1e-12 and np.abs(V.conj().T @ V - np.eye(n)).max() <= 1e-12
        # a well separated spectrum on the same engine: the device solver is used; MAUS_EIGH_TRIDIAG=host switches it off
        B = scenarios.hermitian(n, 77)
        wb = eng.device_eigh(B)
        assert eng.tridiag_solver == "device"
        assert np.abs(wb - sla.eigh(B, eigvals_only=True)).max() <= 1e-13
        monkeypatch.setenv("MAUS_EIGH_TRIDIAG", "host")
        wh = eng.device_eigh(scenarios.hermitian(n, 78))
        assert eng.tridiag_solver == "host" and np.isfinite(wh).all()
    finally:
        eng.ctx.close()


@pytest.mark.parametrize("scale", [1e-140, 1e120])
def test_tridiagonal_solver_is_scale_free(ctx, scale):
    n = 200
    A = scenarios.hermitian(n, 91) * scale
    ctx.set_matrix(A)
    d, e = ctx.herm_tridiag()
    w, (gap, resid, tnorm) = ctx.herm_tridiag_eig(d, e)
    wl = sla.eigh(A, eigvals_only=True)
    assert np.abs(w - wl).max() <= 40 * n * EPS * np.abs(wl).max()
    assert gap > 1e-6 and resid < 1e-13
    ctx.herm_backtransform(None)
    V = ctx.get_eigvecs()
    assert np.abs(A @ V - V * w[None, :]).max() <= 60 * n * EPS * np.abs(wl).max()


def test_hermitian_eigenproblem_takes_its_condition_number_from_the_device_decomposition(monkeypatch):
    """With the decomposition on the device the start-up diagnostics skip the LU-based estimator: the eigenvalues the shortcut
    needs anyway give the exact 2-norm condition number (sigma_i = |lambda_i|), and the first loop body reuses the decomposition."""
    import random
    from adaptive_matrix_solver_amd import engine as eng_mod
    from adaptive_matrix_solver_amd.solver import MAUS_Solver, ProblemType, SolutionCandidate
    n = 1100                                                     # above MAUS_Solver's cond_exact_max
    A = scenarios.hermitian(n, 31)
    exact = np.linalg.cond(A)

    def no_estimator(*a, **k):
        raise AssertionError("the LU-based condition estimator ran")
    monkeypatch.setattr(eng_mod, "estimate_condition_number", no_estimator)
    calls = {"eigh": 0}
    real = eng_mod.DeviceEngine.device_eigh
    monkeypatch.setattr(eng_mod.DeviceEngine, "device_eigh", lambda self, M: (calls.__setitem__("eigh", calls["eigh"] + 1), real(self, M))[1])
    np.random.seed(5); random.seed(5); SolutionCandidate._candidate_id_counter = 0
    s = MAUS_Solver(A, ProblemType.EIGENVALUE, initial_num_candidates=10, quiet=True, eigh_mode="device")
    try:
        assert s.diag_info.get("condition_number_from_eigh") is True and not s.diag_info["condition_number_is_estimate"]
        assert abs(s.cond_number - exact) <= 1e-9 * exact
        assert s.engine.tridiag_solver == "device"
        s.loop_body(1)
        assert calls["eigh"] == 1
        S = SolutionCandidate.State
        assert all(c.state in (S.CONVERGED, S.RETIRED) for c in s.candidates if c.id < 10)
    finally:
        s.engine.ctx.close()


def test_eigenvalues_only_entry_point(ctx):
    """maus_herm_tridiag_eigvals: the bisection alone -- repeated eigenvalues (the +-sigma pairs and zeros of a Hermitian embedding)
    are no obstacle when no vectors are asked for."""
    rng = np.random.default_rng(8)
    B = rng.standard_normal((90, 60)) + 1j * rng.standard_normal((90, 60))
    Hm = np.zeros((150, 150), dtype=np.complex128)
    Hm[:90, 90:] = B
    Hm[90:, :90] = B.conj().T
    ctx.set_matrix(Hm)
    d, e = ctx.herm_tridiag()
    w = ctx.herm_tridiag_eigvals(d, e)
    ref = np.sort(np.concatenate([sla.svd(B, compute_uv=False), -sla.svd(B, compute_uv=False), np.zeros(30)]))
    assert w.shape == (150,) and np.all(np.diff(w) >= 0) and np.abs(w - ref).max() <= 1e-13 * ref[-1] * 150
    one = ctx.herm_tridiag_eigvals(np.array([3.5]), np.zeros(0))
    assert one.tolist() == [3.5]
