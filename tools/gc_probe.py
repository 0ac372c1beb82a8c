"""Is the periodic slow loop body of configs[4] the cyclic garbage collector?  (tools, not product)"""
import gc, os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, scenarios
from adaptive_matrix_solver_amd.solver import MAUS_Solver, ProblemType, SolutionCandidate
A = scenarios.prescribed_svd(2048, 2048, 2048, -8.0)
np.random.seed(1234); random.seed(1234); SolutionCandidate._candidate_id_counter = 0
s = MAUS_Solver(A, ProblemType.SVD, initial_num_candidates=512, quiet=True)
ev = []
t_gc = [0.0]
def cb(phase, info):
    if phase == "start": cb.t = time.perf_counter()
    else:
        dt = time.perf_counter() - cb.t; t_gc[0] += dt
        if info["generation"] == 2: ev.append((round(dt * 1e3, 1), info["collected"]))
gc.callbacks.append(cb)
for mode in ("gc on", "gc off"):
    if mode == "gc off": gc.disable()
    out = []
    for it in range(12):
        t_gc[0] = 0.0; t0 = time.perf_counter(); s.loop_body(it + 1); out.append((round((time.perf_counter() - t0) * 1e3, 1), round(t_gc[0] * 1e3, 1)))
    print(mode, "(body ms, of which gc ms):", out)
print("gen2 collections (ms, collected):", ev, "tracked objects now:", len(gc.get_objects()))
