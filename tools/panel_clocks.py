"""Where a base panel's time goes (debug build only: make EXTRA=-DMAUS_PANEL_CLOCK): wall-clock ticks (10 ns) that thread 0
of every lu_panel_ip_kernel workgroup spent in each phase, summed over workgroups.   python tools/panel_clocks.py [G]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import scenarios  # noqa: E402
from adaptive_matrix_solver_amd import Context, _cabi  # noqa: E402
from adaptive_matrix_solver_amd._cabi import PERT_MT19937  # noqa: E402

if os.environ.get("MAUS_LIB"):                               # the -DMAUS_PANEL_CLOCK build kept beside the shipped library
    _cabi.LIB_PATH = os.environ["MAUS_LIB"]

n = int(os.environ.get("LU_N", 4096))
G = int(sys.argv[1]) if len(sys.argv) > 1 else 32
A = scenarios.ginibre(n, n)
ctx = Context(0)
ctx.set_matrix(A)
ctx.pop_reserve(G)
rng = np.random.default_rng(1)
V = (rng.standard_normal((G, n)) + 1j * rng.standard_normal((G, n))) / np.sqrt(n)
ctx.pop_put(0, list(range(G)), V)
num, den = ctx.matvec_rayleigh(list(range(G)))
lam = num / den
psi = np.full(G, 1e-20)
np.random.seed(3)
st = np.random.get_state()
ctx.lu_reserve(n, G)
desc = (st, 4 * n * n, 0, np.arange(G, dtype=np.int32))
sl = list(range(G))
ctx.shifted_lu_solve(sl, lam, psi, 0, PERT_MT19937, desc)
lib = _cabi.load_library()
out = (ctypes.c_ulonglong * 16)()
lib.maus_debug_panel_clocks(out, 1)
ctx.shifted_lu_solve(sl, lam, psi, 0, PERT_MT19937, desc)
lib.maus_debug_panel_clocks(out, 1)
names = ["prologue (perm load)", "(b') pivot-row block + solve", "(c') load + left-looking update", "column loop", "store + barrier", "rs: search + reduce + barrier", "rs: pick + publish + barrier", "rs: interchange + update",
         "mw: load panel slice", "mw: scan, reduce, stage candidate", "mw: publish + drain", "mw: arrive + wait", "mw: read candidates", "mw: pick winner",
         "mw: interchange + update", "mw: store slice"]
tot = sum(out)
print('(mw rows: summed over the W workgroups of a matrix)')
print(f"G={G}: per matrix and factorisation (256 panels), ms of thread 0's wall clock; total {tot * 1e-5 / G:.2f} ms")
for nm, v in zip(names, out):
    if v:
        print(f"  {nm:34s} {v * 1e-5 / G:8.3f} ms  ({100.0 * v / tot:4.1f} %)   {v * 1e-2 / G / (n // 16):7.2f} us per panel")
ctx.close()
