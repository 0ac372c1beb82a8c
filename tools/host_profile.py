"""cProfile of loop bodies of a BASELINE configuration on the device (tools, not product): where the HOST time of a loop
body goes once the kernels are fast.

    python tools/host_profile.py c2|c3|c4|c5 [loop bodies]        (c4: the profile starts at the FIRST loop body)
"""
import cProfile, pstats, os, sys, random, io, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, scenarios
from adaptive_matrix_solver_amd.solver import MAUS_Solver, ProblemType, SolutionCandidate
cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
bodies = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if cfg == "c5":
    A, PT, P = scenarios.prescribed_svd(2048, 2048, 2048, -8.0), ProblemType.SVD, 512
elif cfg == "c2":
    A, PT, P = scenarios.ginibre(1024, 1024), ProblemType.EIGENVALUE, 256
elif cfg == "c3":
    (A, bvec), PT, P = scenarios.wide_diag_system(4096, 4096, decades=7.0, offdiag=0.1), ProblemType.SOLVE_LINEAR_SYSTEM, 512
elif cfg == "c4":
    A, PT, P = scenarios.hermitian(int(os.environ.get("C4_N", "8192")), 8192), ProblemType.EIGENVALUE, 128
else:
    raise SystemExit("c2, c3, c4 or c5")
np.random.seed(1234); random.seed(1234); SolutionCandidate._candidate_id_counter = 0
s = MAUS_Solver(A, PT, b_vector=bvec if cfg == "c3" else None, initial_num_candidates=P, quiet=True)
S_ = SolutionCandidate.State


def body(it):
    if cfg == "c3":                              # Jacobi is built only for stuck_counter > 1 (AMS:65); as bench.py --config c3
        for c in s.candidates:
            if c.state not in (S_.CONVERGED, S_.RETIRED):
                c.stuck_counter = 2
    return s.loop_body(it)


if os.environ.get("SPLIT_SYNC"):                 # bill pending device work to a sync in front of every read-back, not to the read-back
    _ctx = s.engine.ctx; _pg = _ctx.pop_get; _sy = _ctx.sync
    def sync_before_pop_get(): _sy()
    def pop_get_split(*a, **k):
        sync_before_pop_get()
        return _pg(*a, **k)
    _ctx.pop_get = pop_get_split
    for _nm in ("hist_append", "herm_match", "residual", "pop_put", "gram"):
        def _mk(nm, real):
            def f(*a, **k):
                r = real(*a, **k); t0 = time.perf_counter(); _sy(); dt = (time.perf_counter() - t0) * 1e3
                if dt > 0.5: print(f"[split] device work left behind by {nm}: {dt:.2f} ms")
                return r
            return f
        setattr(_ctx, _nm, _mk(_nm, getattr(_ctx, _nm)))
if cfg != "c4":
    body(1)
s.engine.ctx.sync()
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
act = 0
for it in range(bodies):
    act += body((1 if cfg == "c4" else 2) + it)
s.engine.ctx.sync()
pr.disable(); el = time.perf_counter() - t0
print(f"{cfg}: {act} candidate steps in {bodies} loop bodies, {el * 1e3:.1f} ms, {el / act * 1e6:.1f} us per candidate step")
out = io.StringIO(); st = pstats.Stats(pr, stream=out); st.sort_stats("tottime").print_stats(32)
print("\n".join(l[:170] for l in out.getvalue().splitlines()[:48]))
if len(sys.argv) > 3:                                      # callees of one function, e.g. _solve
    out = io.StringIO(); st.stream = out; st.print_callees(sys.argv[3])
    print("\n".join(l[:170] for l in out.getvalue().splitlines()[:60]))
