"""cProfile of MAUS_Solver construction for BASELINE configs[3] (tools, not product).   python tools/build_profile.py [n]"""
import cProfile, io, os, pstats, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, scenarios
from adaptive_matrix_solver_amd.solver import MAUS_Solver, ProblemType, SolutionCandidate
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
A = scenarios.hermitian(n, n)
np.random.seed(1234); random.seed(1234); SolutionCandidate._candidate_id_counter = 0
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
s = MAUS_Solver(A, ProblemType.EIGENVALUE, initial_num_candidates=128, quiet=True)
pr.disable(); print(f"construction {time.perf_counter() - t0:.2f} s")
out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(18)
print("\n".join(l[:160] for l in out.getvalue().splitlines()[:34]))
