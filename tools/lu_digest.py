"""SHA-256 of the solutions and pivot sequences of fixed LU batches: a bitwise before / after check for changes to the panel
kernels that must not change any result (tools, not product).   python tools/lu_digest.py"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import scenarios  # noqa: E402
from adaptive_matrix_solver_amd import Context, _cabi  # noqa: E402
if os.environ.get("MAUS_LIB"):                               # an older build of the library, for the "before" digests
    _cabi.LIB_PATH = os.environ["MAUS_LIB"]
from adaptive_matrix_solver_amd._cabi import PERT_NONE  # noqa: E402

for n, G in ((4096, 8), (4096, 40), (4096, 160), (1024, 64), (2048, 24), (300, 5)):
    A = scenarios.ginibre(n, n)
    ctx = Context(0)
    ctx.set_matrix(A)
    ctx.pop_reserve(G)
    rng = np.random.default_rng(n + G)
    V = (rng.standard_normal((G, n)) + 1j * rng.standard_normal((G, n))) / np.sqrt(n)
    sl = list(range(G))
    ctx.pop_put(0, sl, V)
    lam = (rng.standard_normal(G) + 1j * rng.standard_normal(G)) * 0.5
    st = ctx.shifted_lu_solve(sl, lam, np.full(G, 1e-20), 0, PERT_NONE, None)
    W = ctx.pop_get(2, sl, n)
    print(f"n={n} G={G}: status {int(np.abs(st).sum())} W {hashlib.sha256(W.tobytes()).hexdigest()[:24]}", flush=True)
    ctx.close()
rng = np.random.default_rng(50)
n = 5000
A = (rng.standard_normal((1, n, n)) + 1j * rng.standard_normal((1, n, n))) / np.sqrt(n)
b = rng.standard_normal((1, n)) + 1j * rng.standard_normal((1, n))
ctx = Context(0)
x, status, ipiv = ctx.lu_solve(A, b, want_ipiv=True)
print(f"n=5000 G=1: x {hashlib.sha256(x.tobytes()).hexdigest()[:24]} ipiv {hashlib.sha256(ipiv.tobytes()).hexdigest()[:24]}")
ctx.close()
