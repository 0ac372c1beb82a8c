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
