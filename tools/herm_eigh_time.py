"""Where the device Hermitian eigendecomposition spends its time (csrc/herm.hip): tridiagonalisation on the device, dstemr on
the host, back-transformation on the device; checked against A V = V diag(w).    python tools/herm_eigh_time.py [n ...]"""
import os
import sys
import time

import numpy as np
import scipy.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import scenarios  # noqa: E402
from adaptive_matrix_solver_amd import Context  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [2048, 4096, 8192]:
    A = scenarios.hermitian(n, 8192)
    ctx = Context(0)
    ctx.set_matrix(A)
    t0 = time.perf_counter(); d, e = ctx.herm_tridiag(); t1 = time.perf_counter()
    if os.environ.get("HERM_TRIDIAG", "device") == "host":
        w, Z = sla.eigh_tridiagonal(d, e); t2 = time.perf_counter()
        ctx.herm_backtransform(Z); t3 = time.perf_counter()
        what = "dstemr (host)"
    else:
        w, diag = ctx.herm_tridiag_eig(d, e); t2 = time.perf_counter()
        ctx.herm_backtransform(None); t3 = time.perf_counter()
        what = f"bisection + twisted factorisation (device; min gap / ||T|| {diag[0]:.1e}, max residual / ||T|| {diag[1]:.1e})"
    line = f"n={n}: tridiag (device) {t1 - t0:.2f} s, {what} {t2 - t1:.2f} s, back-transform (device) {t3 - t2:.2f} s, total {t3 - t0:.2f} s"
    # residual on the device-resident V through a few columns
    V = ctx.get_eigvecs()
    k = np.linspace(0, n - 1, 16).astype(int)
    res = np.linalg.norm(A @ V[:, k] - V[:, k] * w[k][None, :], axis=0).max()
    orth = np.abs(V[:, k].conj().T @ V[:, k] - np.eye(len(k))).max()
    kk = np.arange(min(n, 512)); orth = max(orth, np.abs(V[:, kk].conj().T @ V[:, kk] - np.eye(len(kk))).max())      # neighbours too
    line += f"; max ||A v - w v|| over 16 columns {res:.2e}, orthogonality {orth:.2e}, Im V[0] {np.abs(V[0].imag).max():.1e}"
    if n <= 4096 or os.environ.get("HERM_HOST"):
        t4 = time.perf_counter(); wl = sla.eigh(A, eigvals_only=False)[0]; t5 = time.perf_counter()
        line += f"; host scipy.linalg.eigh {t5 - t4:.1f} s, max |w - w_lapack| {np.abs(w - wl).max():.2e}"
    print(line, flush=True)
    ctx.close()
