"""ms per solve of one maus_shifted_lu_solve call (n = 4096 or LU_N, device-regenerated perturbation: the bench's mode) as a function
of the batch size -- the per-rank workload of a population sharded over N GPUs is pop/N solves per step.

    python tools/lu_batch_rates.py [G ...]        (environment switches of the library apply: MAUS_PANEL_MW, MAUS_LU_STREAMS)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import scenarios  # noqa: E402
from adaptive_matrix_solver_amd import Context  # noqa: E402
from adaptive_matrix_solver_amd._cabi import PERT_MT19937  # noqa: E402

n = int(os.environ.get("LU_N", 4096))
sizes = [int(a) for a in sys.argv[1:]] or [8, 16, 32, 48, 64, 96, 128, 192, 256, 331]
A = scenarios.ginibre(n, n)
ctx = Context(0)
ctx.set_matrix(A)
P = max(sizes)
ctx.pop_reserve(P)
rng = np.random.default_rng(1)
V = (rng.standard_normal((P, n)) + 1j * rng.standard_normal((P, n))) / np.sqrt(n)
ctx.pop_put(0, list(range(P)), V)
num, den = ctx.matvec_rayleigh(list(range(P)))
lam = num / den
psi = np.full(P, 1e-20)
np.random.seed(3)
st = np.random.get_state()
cap = ctx.lu_reserve(n, int(os.environ.get("LU_RESERVE", P)))
print(f"# MAUS_PANEL_MW={os.environ.get('MAUS_PANEL_MW', '1')} MAUS_LU_STREAMS={os.environ.get('MAUS_LU_STREAMS', '1')} workspace capacity {cap}")
for G in sizes:
    sl = list(range(G))
    desc = (st, 4 * n * n, 0, np.arange(G, dtype=np.int32))
    ctx.shifted_lu_solve(sl, lam[:G], psi[:G], 0, PERT_MT19937, desc)          # warm
    reps = 3 if G <= 64 else 2
    t0 = time.perf_counter()
    for _ in range(reps):
        status = ctx.shifted_lu_solve(sl, lam[:G], psi[:G], 0, PERT_MT19937, desc)
    dt = (time.perf_counter() - t0) / reps
    assert (status == 0).all()
    line = f"G={G:4d}: {dt * 1e3:8.1f} ms per call, {dt * 1e3 / G:6.3f} ms per solve, {G / dt:7.1f} solves/s"
    if os.environ.get("LU_BATCH_KERNELS"):            # one more call with every launch bracketed by HIP events
        ctx.profile_enable(1)
        ctx.shifted_lu_solve(sl, lam[:G], psi[:G], 0, PERT_MT19937, desc)
        pr = ctx.profile_read()
        ctx.profile_enable(False)
        line += " | " + " ".join(f"{k}={v['ms']:.1f}" for k, v in pr.items() if v["ms"] > 0.05)
    print(line, flush=True)
ctx.close()
