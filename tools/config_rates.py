"""Candidate-step rates of the BASELINE.json parity configurations on one GPU (not the bench metric).

    python tools/config_rates.py [c1] [c2] [c3] [c4]        (default: all)

Each configuration builds the solver through the reference-compatible API, runs one warm-up loop body and
two timed ones (AMS:573-577 per iteration), and prints one line.  Sizes are BASELINE.json's except where
noted in the output (a single GPU's share of the sharded configurations)."""
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import scenarios  # noqa: E402
from adaptive_matrix_solver_amd.solver import MAUS_Solver, ProblemType, SolutionCandidate  # noqa: E402


def run(tag, A, kind, pop, b=None, note=""):
    np.random.seed(1234)
    random.seed(1234)
    SolutionCandidate._candidate_id_counter = 0
    t0 = time.perf_counter()
    solver = MAUS_Solver(A, kind, b_vector=b, initial_num_candidates=pop, global_convergence_tol=1e-8, quiet=True,
                         record_history=False)
    t_build = time.perf_counter() - t0
    ctx = solver.engine.ctx
    solver.loop_body(1)
    ctx.sync()
    t0 = time.perf_counter()
    steps = 0
    for it in (2, 3):
        steps += solver.loop_body(it)
    ctx.sync()
    el = time.perf_counter() - t0
    pref = solver.problem_knowledge.get("local_solver_preference")
    print(f"{tag}: {steps / el:9.1f} candidate-steps/s  ({steps} steps in {el * 1e3:.0f} ms, solver build {t_build:.1f} s, "
          f"preferred solver {pref}) {note}", flush=True)


which = set(sys.argv[1:]) or {"c1", "c2", "c3", "c4"}
if "c1" in which:
    run("configs[1] 1024x1024 non-Hermitian eig, 256 candidates", scenarios.ginibre(1024, 1024), ProblemType.EIGENVALUE, 256)
if "c2" in which:
    A, b = scenarios.wide_diag_system(4096, 4096, decades=7.0, offdiag=0.1)
    run("configs[2] 4096x4096 linear system, 512 candidates (GMRES+Jacobi preferred)", A, ProblemType.SOLVE_LINEAR_SYSTEM, 512, b=b)
if "c4" in which:
    run("configs[4] 2048x2048 SVD cond 1e8, 64 candidates", scenarios.prescribed_svd(2048, 2048, 77, -8.0), ProblemType.SVD, 64,
        note="[one GPU's share of 512]")
if "c3" in which:
    run("configs[3] Hermitian eig at 4096x4096, 128 candidates", scenarios.hermitian(4096, 8192), ProblemType.EIGENVALUE, 128,
        note="[one GPU's share of 1024; n=4096 instead of 8192 so that the host eigh (once per matrix) stays under a minute]")
