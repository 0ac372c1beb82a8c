"""Loop-body time of a Hermitian run (every candidate converges in the first step, so later loop bodies are the
distinctness / redundancy tests of AMS:424-475 and 504-549) with and without the device Gram block (SURVEY f-2)."""
import os, random, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import scenarios
from adaptive_matrix_solver_amd.solver import MAUS_Solver, ProblemType, SolutionCandidate

n, P = int(os.environ.get("N", 4096)), int(os.environ.get("P", 1024))
A = scenarios.hermitian(n, n)
for gram_min in (8, 10 ** 9):
    np.random.seed(1); random.seed(1); SolutionCandidate._candidate_id_counter = 0
    s = MAUS_Solver(A, ProblemType.EIGENVALUE, initial_num_candidates=P, quiet=True, record_history=False, gram_min=gram_min)
    ts = []
    for it in range(1, 5):
        t = time.perf_counter(); s.loop_body(it); s.engine.ctx.sync(); ts.append(time.perf_counter() - t)
    print(f"n={n} P={P} gram_min={gram_min}: loop bodies {[round(x * 1e3, 1) for x in ts]} ms, "
          f"distinct converged {s.num_distinct_converged_solutions}, population {len(s.candidates)}", flush=True)
