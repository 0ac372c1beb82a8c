"""Per loop body of configs[1]: wall time, time inside the cyclic garbage collector, time inside maus_shifted_lu_solve -- to place
a one-off slow body (tools, not product).   python tools/body_probe.py [bodies] [c2|c3|c5]"""
import gc, os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, scenarios
from adaptive_matrix_solver_amd.solver import MAUS_Solver, ProblemType, SolutionCandidate
cfg = sys.argv[2] if len(sys.argv) > 2 else "c2"
bvec = None
if cfg == "c5":
    A, PT, P = scenarios.prescribed_svd(2048, 2048, 2048, -8.0), ProblemType.SVD, 512
elif cfg == "c3":
    (A, bvec), PT, P = scenarios.wide_diag_system(4096, 4096, decades=7.0, offdiag=0.1), ProblemType.SOLVE_LINEAR_SYSTEM, 512
else:
    A, PT, P = scenarios.ginibre(1024, 1024), ProblemType.EIGENVALUE, 256
np.random.seed(1234); random.seed(1234); SolutionCandidate._candidate_id_counter = 0
s = MAUS_Solver(A, PT, b_vector=bvec, initial_num_candidates=P, quiet=True)
S_ = SolutionCandidate.State
ctx = s.engine.ctx
import traceback
t_gc = [0.0]; t_lu = [0.0]; t_other = {}; calls = []
def cb(phase, info):
    if phase == "start": cb.t = time.perf_counter()
    else: t_gc[0] += time.perf_counter() - cb.t
gc.callbacks.append(cb)
def wrap(name):
    real = getattr(ctx, name)
    def f(*a, **k):
        t0 = time.perf_counter(); r = real(*a, **k); dt = time.perf_counter() - t0
        if name == "shifted_lu_solve": t_lu[0] += dt
        else: t_other[name] = t_other.get(name, 0.0) + dt
        if name == "pop_get":
            calls.append((len(a[1]), round(dt * 1e3, 2), "".join(traceback.format_stack(limit=6)[-5:-1]).count("_bulk_pull")))
        return r
    setattr(ctx, name, f)
for nm in ("shifted_lu_solve", "hist_append", "pop_put", "pop_get", "matvec_rayleigh", "residual", "relax_normalise", "lu_reserve", "pop_reserve", "gram", "gmres", "gmres_pert", "svd_power_propose", "svd_commit", "hist_get", "pop_copy", "matmul", "linear_residual"):
    if hasattr(ctx, nm): wrap(nm)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 11):
    t_gc[0] = 0.0; t_lu[0] = 0.0; t_other.clear(); calls.clear()
    if cfg == "c3":
        for c in s.candidates:
            if c.state not in (S_.CONVERGED, S_.RETIRED):
                c.stuck_counter = 2
    t0 = time.perf_counter(); act = s.loop_body(it + 1); ctx.sync(); el = time.perf_counter() - t0
    print(f"body {it + 1}: {act} active, {el * 1e3:.1f} ms, gc {t_gc[0] * 1e3:.1f}, lu call {t_lu[0] * 1e3:.1f}, others " +
          ", ".join(f"{k} {v * 1e3:.1f}" for k, v in sorted(t_other.items(), key=lambda kv: -kv[1])[:5]) +
          (f" | pop_get (rows, ms, bulk): {calls[:6]} ... {len(calls)} calls" if calls else ""))
