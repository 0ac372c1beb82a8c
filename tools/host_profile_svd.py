"""cProfile of two loop bodies of the 2048 x 2048 SVD configuration (tools, not product)."""
import cProfile, pstats, os, sys, random, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, scenarios
from adaptive_matrix_solver_amd.solver import MAUS_Solver, ProblemType, SolutionCandidate
A = scenarios.prescribed_svd(2048, 2048, 77, -8.0)
np.random.seed(1234); random.seed(1234); SolutionCandidate._candidate_id_counter = 0
s = MAUS_Solver(A, ProblemType.SVD, initial_num_candidates=64, quiet=True, record_history=False)
s.loop_body(1)
pr = cProfile.Profile(); pr.enable()
s.loop_body(2); s.loop_body(3)
pr.disable()
out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(30)
print("\n".join(l[:160] for l in out.getvalue().splitlines()[:50]))
