"""cProfile of the host side of one loop body at the metric configuration (tools, not product)."""
import cProfile, pstats, os, sys, random, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, scenarios
from adaptive_matrix_solver_amd.solver import MAUS_Solver, ProblemType, SolutionCandidate
n, P = int(os.environ.get("N", 4096)), int(os.environ.get("P", 256))
A = scenarios.ginibre(n, n)
np.random.seed(1234); random.seed(1234); SolutionCandidate._candidate_id_counter = 0
s = MAUS_Solver(A, ProblemType.EIGENVALUE, initial_num_candidates=P, quiet=True, record_history=False)
W = int(os.environ.get("WARM", 1))
for it in range(1, W + 1):
    s.loop_body(it)
pr = cProfile.Profile(); pr.enable()
s.loop_body(W + 1); s.loop_body(W + 2)
pr.disable()
out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(28)
print("\n".join(l[:150] for l in out.getvalue().splitlines()[:60]))
