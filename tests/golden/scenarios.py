"""Seeded synthetic inputs shared by the golden generator (make_goldens.py, which
drives the *reference*) and by the tests (which drive the oracle and the HIP
path).  Pure NumPy; nothing here touches /root/reference.

All matrices come from np.random.default_rng(seed) -- never from the legacy
global streams, which the harness seeds afterwards."""
import numpy as np


def ginibre(n, seed, scale=None):
    """Complex Ginibre (G1 + i G2); scale=None -> 1/sqrt(n) (spectrum in the unit disk)."""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    s = (1.0 / np.sqrt(n)) if scale is None else scale
    return (A * s).astype(np.complex128)


def hermitian(n, seed):
    B = ginibre(n, seed)
    return ((B + B.conj().T) / 2.0).astype(np.complex128)


def wide_diag_system(n, seed, decades=7.0, offdiag=0.1):
    """C3-style linear system: diag(10^U(0,decades) e^{2 pi i U}) + offdiag*Ginibre/sqrt(n)."""
    rng = np.random.default_rng(seed)
    d = 10.0 ** rng.uniform(0.0, decades, n) * np.exp(2j * np.pi * rng.uniform(0, 1, n))
    G = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) / np.sqrt(n)
    A = np.diag(d) + offdiag * G
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    return A.astype(np.complex128), b.astype(np.complex128)


def prescribed_svd(m, n, seed, lo=-8.0):
    """C5-style: U diag(logspace(0, lo, min(m,n))) V^H with Haar-ish U, V."""
    rng = np.random.default_rng(seed)
    k = min(m, n)
    U, _ = np.linalg.qr(rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m)))
    V, _ = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    S = np.zeros((m, n), dtype=np.complex128)
    S[np.arange(k), np.arange(k)] = np.logspace(0.0, lo, k)
    return (U @ S @ V.conj().T).astype(np.complex128)


def laplace_like(n, seed, make_hermitian=False):
    """Small structured eig problem in the spirit of the reference's N=8 demo
    (AMS:619-628): tridiagonal -2/1/1 plus a few complex couplings and 1e-3 noise.
    Own generator (default_rng), not the reference's (which uses the global stream)."""
    rng = np.random.default_rng(seed)
    M = np.zeros((n, n), dtype=np.complex128)
    i = np.arange(n)
    M[i, i] = -2.0
    M[i[:-1], i[:-1] + 1] = 1.0
    M[i[:-1] + 1, i[:-1]] = 1.0
    M[0, 2] = 0.5
    M[2, 0] = 0.5j
    M[n // 2 - 1, n // 2] = 1.5 + 0.5j
    M[n // 2, n // 2 - 1] = -1.5 + 0.5j
    M += (rng.uniform(-1, 1, (n, n)) + 1j * rng.uniform(-1, 1, (n, n))) * 1e-3
    if make_hermitian:
        M = (M + M.conj().T) / 2.0
    return M


# name -> dict(kind, matrix builder args, population, iterations, harness seed)
TRAJECTORIES = {
    # G3 / C1: dense non-Hermitian eig, direct LU path
    "eig16":   dict(kind="eig", build=("ginibre", 16, 16, 1.0), P=16, iters=10, seed=1234, tol=1e-8),
    "eig64":   dict(kind="eig", build=("ginibre", 64, 64, 1.0), P=16, iters=10, seed=1234, tol=1e-8),
    "eig48u":  dict(kind="eig", build=("ginibre", 48, 48, None), P=12, iters=12, seed=99, tol=1e-8),
    # G4: small structured problems with convergence / retire / spawn events
    "lap8":    dict(kind="eig", build=("laplace", 8, 8, False), P=30, iters=60, seed=7, tol=1e-7),
    "lap8h":   dict(kind="eig", build=("laplace", 8, 8, True), P=30, iters=6, seed=7, tol=1e-7),
    # G6: Hermitian shortcut and SVD power steps
    "herm16":  dict(kind="eig", build=("hermitian", 16, 16), P=12, iters=3, seed=5, tol=1e-8),
    "herm64":  dict(kind="eig", build=("hermitian", 64, 64), P=16, iters=3, seed=5, tol=1e-8),
    "svd5x4":  dict(kind="svd", build=("svd", 5, 4, 21, -3.0), P=25, iters=30, seed=11, tol=1e-6),
    "svd64":   dict(kind="svd", build=("svd", 64, 48, 22, -8.0), P=40, iters=20, seed=11, tol=1e-8),
    # AMS:243-247: A = 1e-10 * (a dense unitary) puts ||A v|| and ||A^H u|| on the 1e-10 knife edge of AMS:235 / AMS:242,
    # so that the tiny-sigma convergence branch is taken with a right vector that was left un-normalised (norm < 1e-10)
    # and is replaced by ones/sqrt(n) (AMS:247: 18 times in this run); the collapse branches AMS:236-239 fire too (17 times)
    "svdtiny": dict(kind="svd", build=("scaled_unitary", 6, 3, 1e-10), P=12, iters=4, seed=1, tol=1e-8),
    # G7: linear systems -- stable/direct, and Fragile (GMRES preferred -> LU fallback under SciPy>=1.14)
    "lin24":   dict(kind="lin", build=("ginibre_b", 24, 24), P=10, iters=12, seed=3, tol=1e-8),
    "lin32f":  dict(kind="lin", build=("widediag", 32, 32, 7.0), P=10, iters=8, seed=3, tol=1e-8),
}


def build(spec):
    """-> (matrix, b or None)"""
    b = spec["build"]
    if b[0] == "ginibre":
        return ginibre(b[1], b[2], b[3]), None
    if b[0] == "ginibre_b":
        A = ginibre(b[1], b[2], 1.0)
        rng = np.random.default_rng(b[2] + 1000)
        return A, (rng.standard_normal(b[1]) + 1j * rng.standard_normal(b[1]))
    if b[0] == "laplace":
        return laplace_like(b[1], b[2], b[3]), None
    if b[0] == "hermitian":
        return hermitian(b[1], b[2]), None
    if b[0] == "svd":
        return prescribed_svd(b[1], b[2], b[3], b[4]), None
    if b[0] == "widediag":
        return wide_diag_system(b[1], b[2], b[3])
    if b[0] == "scaled_unitary":
        rng = np.random.default_rng(b[2])
        Q, _ = np.linalg.qr(rng.standard_normal((b[1], b[1])) + 1j * rng.standard_normal((b[1], b[1])))
        return (b[3] * Q).astype(np.complex128), None
    raise KeyError(b[0])


def solve_case_inputs(key):
    """Inputs of the InverseIterateSolver.solve fixtures (solve_cases.json/npz):
    -> (A_target, rhs).  Rebuilt on both sides so the fixture stores outputs only."""
    if key.startswith("direct_n"):
        n = int(key.split("_")[1][1:])
        A = ginibre(n, 100 + n, 1.0)
        rng = np.random.default_rng(200 + n)
        rhs = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        shift = complex(rng.standard_normal(), rng.standard_normal())
        return A - shift * np.eye(n), rhs
    if key == "bigpsi":
        A = ginibre(16, 777, 1.0)
        rng = np.random.default_rng(778)
        return A, rng.standard_normal(16) + 1j * rng.standard_normal(16)
    if key.startswith("gmres_n"):
        n = int(key.split("_")[1][1:])
        return wide_diag_system(n, 300 + n, decades=3.0)
    if key.startswith("gmresbig_n"):
        n = int(key.split("_")[1][1:])
        return wide_diag_system(n, 400 + n, decades=3.0)
    if key.startswith("gmresfb_n"):
        n = int(key.split("_")[1][1:])
        return wide_diag_system(n, 300 + n, decades=7.0)
    if key == "gmres_legacy":
        return wide_diag_system(32, 332, decades=3.0)
    raise KeyError(key)
