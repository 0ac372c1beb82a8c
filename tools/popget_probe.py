"""First-call costs of maus_pop_get (tools, not product)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, scenarios
from adaptive_matrix_solver_amd import Context
n = 1024
ctx = Context(0); ctx.set_matrix(scenarios.ginibre(n, 1)); ctx.pop_reserve(600)
def t(label, f):
    t0 = time.perf_counter(); f(); print(f"{label}: {(time.perf_counter() - t0) * 1e3:.2f} ms")
t("gram 3 (small scratch first)", lambda: ctx.gram(0, [1, 5, 9], n))
for rep in range(2):
    t("1 row", lambda: ctx.pop_get(0, [5], n))
    t("3 scattered rows", lambda: ctx.pop_get(0, [5, 9, 300], n))
    t("11 scattered rows", lambda: ctx.pop_get(0, list(range(3, 300, 28)), n))
    t("200 scattered rows", lambda: ctx.pop_get(0, list(range(1, 600, 3)), n))
    t("64 contiguous rows", lambda: ctx.pop_get(0, list(range(64)), n))
    t("gram 11", lambda: ctx.gram(0, list(range(3, 300, 28)), n))
# large read-backs (n = 8192: 128 vectors = 16 MB) and their correctness
n2 = 8192
ctx2 = Context(0); A2 = scenarios.ginibre(n2, 2); ctx2.set_matrix(A2); ctx2.pop_reserve(200)
rng = np.random.default_rng(0); X = rng.standard_normal((150, n2)) + 1j * rng.standard_normal((150, n2))
ctx2.pop_put(0, list(range(150)), X)
for rep in range(2):
    for sl in (list(range(128)), list(range(1, 150, 2)) + [0, 4], list(range(150))):
        t0 = time.perf_counter(); Y = ctx2.pop_get(0, sl, n2); dt = (time.perf_counter() - t0) * 1e3
        print(f"n = 8192, {len(sl)} rows: {dt:.2f} ms, equal: {np.array_equal(Y, X[sl])}")
Y = ctx2.pop_get(0, [3, 7, 100], 5000); print("partial length:", np.array_equal(Y, X[[3, 7, 100], :5000]))
