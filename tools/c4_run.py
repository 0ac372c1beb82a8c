"""BASELINE.json configs[3] end to end on one GPU: 8192 x 8192 Hermitian eigenproblem through MAUS_Solver, one GPU's
share (128) of the 1024 candidates.  Prints where the time goes: solver construction (symmetry checks + condition
estimate), the once-per-matrix host eigh (scipy.linalg.eigh -- the reference's own call, AMS:161, made ONCE instead of once
per candidate per step) and the device loop bodies (similarity GEMM conj(X) V, arg-max pick, residual GEMM).

    python tools/c4_run.py [n] [candidates]
"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
P = int(sys.argv[2]) if len(sys.argv) > 2 else 128
try:
    from threadpoolctl import threadpool_limits
    threadpool_limits(limits=int(os.environ.get("C4_THREADS", "16")))
except Exception:
    pass
t0 = time.perf_counter()
A = scenarios.hermitian(n, 8192)
print(f"matrix {n}x{n} Hermitian ((B+B^H)/2, B Ginibre/sqrt(n)) built in {time.perf_counter() - t0:.1f} s", flush=True)
import scipy.linalg as sla
_eigh = sla.eigh
t_eigh = [0.0]


def timed_eigh(*a, **k):
    t = time.perf_counter()
    r = _eigh(*a, **k)
    t_eigh[0] += time.perf_counter() - t
    return r


sla.eigh = timed_eigh
np.random.seed(1234); random.seed(1234); SolutionCandidate._candidate_id_counter = 0
t0 = time.perf_counter()
solver = MAUS_Solver(A, ProblemType.EIGENVALUE, initial_num_candidates=P, global_convergence_tol=1e-8, quiet=True, record_history=False)
t_build = time.perf_counter() - t0
print(f"solver construction: {t_build:.1f} s, of which host eigh {t_eigh[0]:.1f} s (Hermitian={solver.problem_knowledge['is_hermitian']}, "
      f"cond={solver.cond_number:.3e}, estimate={solver.diag_info.get('condition_number_is_estimate')}, "
      f"from eigh={solver.diag_info.get('condition_number_from_eigh', False)})", flush=True)
eigh_at_build = t_eigh[0]
rows = []
for it in range(1, 4):
    t0 = time.perf_counter()
    steps = solver.loop_body(it)
    solver.engine.ctx.sync()
    dt = time.perf_counter() - t0
    rows.append((it, steps, dt, t_eigh[0]))
    print(f"loop body {it}: {steps} candidate steps in {dt:.3f} s (host eigh so far {t_eigh[0]:.1f} s), "
          f"distinct converged {solver.num_distinct_converged_solutions}, population {len(solver.candidates)}", flush=True)
first = rows[0]
later = rows[1:]
print(f"SUMMARY n={n} P={P}: host eigh (once per matrix) {t_eigh[0]:.1f} s; first loop body without it {first[2] - (first[3] - eigh_at_build):.3f} s; "
      f"later loop bodies {', '.join(f'{r[2] * 1e3:.1f} ms / {r[1]} steps' for r in later)}; "
      f"the reference would call eigh {first[1]} times in the first iteration alone ({first[1] * t_eigh[0] / 3600:.1f} h at this speed)")
c = solver.candidates[0]
v = np.asarray(c.v_k)
print(f"check: candidate 0 residual reported {c.residual_k:.3e}, recomputed {np.linalg.norm(A @ v - c.lambda_k * v):.3e}")
