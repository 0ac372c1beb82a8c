"""Why is the first loop body of a fresh configs[2] population slow?  Per body: wall, gmres calls / candidates / inner iterations, kernel ms by class."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, scenarios
from adaptive_matrix_solver_amd.solver import MAUS_Solver, ProblemType, SolutionCandidate
(A, b) = scenarios.wide_diag_system(4096, 4096, decades=7.0, offdiag=0.1)
np.random.seed(1234); random.seed(1234); SolutionCandidate._candidate_id_counter = 0
s = MAUS_Solver(A, ProblemType.SOLVE_LINEAR_SYSTEM, b_vector=b, initial_num_candidates=512, quiet=True)
ctx = s.engine.ctx
S_ = SolutionCandidate.State
calls = []
_g = ctx.gmres
def g(*a, **k):
    t = time.perf_counter(); r = _g(*a, **k); calls.append((len(r[1]), int(np.sum(r[1])), int(np.max(r[1])), round((time.perf_counter() - t) * 1e3, 1))); return r
ctx.gmres = g
_lu = ctx.shifted_lu_solve
def lu(*a, **k):
    t = time.perf_counter(); r = _lu(*a, **k); calls.append(("LU", len(r), round((time.perf_counter() - t) * 1e3, 1))); return r
ctx.shifted_lu_solve = lu
for it in range(4):
    for c in s.candidates:
        if c.state not in (S_.CONVERGED, S_.RETIRED): c.stuck_counter = 2
    calls.clear(); ctx.profile_enable(1); t0 = time.perf_counter(); act = s.loop_body(it + 1); ctx.sync(); dt = time.perf_counter() - t0
    pr = ctx.profile_read(); ctx.profile_enable(False)
    print(f"body {it + 1}: {act} active, {dt * 1e3:.1f} ms; calls (cands, inner sum, inner max, ms): {calls}; kernels: " + " ".join(f"{k}={v['ms']:.1f}/{v['launches']}" for k, v in pr.items() if v["ms"] > 0.05), flush=True)
