import cProfile, pstats, io, os, sys, random
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
import numpy as np, scenarios
from adaptive_matrix_solver_amd.solver import MAUS_Solver, ProblemType, SolutionCandidate
A = scenarios.ginibre(4096, 4096)
np.random.seed(1); random.seed(1)
pr = cProfile.Profile(); pr.enable()
s = MAUS_Solver(A, ProblemType.EIGENVALUE, initial_num_candidates=256, quiet=True, record_history=False)
pr.disable()
out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(22)
print("\n".join(l[:140] for l in out.getvalue().splitlines()[:45]))
