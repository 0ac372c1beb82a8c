#!/usr/bin/env python3
"""bench.py -- candidate-steps/sec of the MAUS hot path on MI355X (BASELINE.json metric).

A "step" is one loop-body iteration of MAUS_Solver (AMS:573-577: diagnostics, strategy, the
batched update_solution_step of every active candidate, population management) on the
metric's configuration: n=4096 dense non-Hermitian eigenproblem, initial_num_candidates=256,
direct-LU path.  `value` = candidate steps executed / wall time of the timed loop bodies, with
A and the population already resident in HBM when the timed region starts.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus 8 ...        # starts 8 ranks itself (torch.distributed.run), or is started by
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line: the metric, `per_step` (wall ms, active candidates and the K>=256 zgemm time of
every timed loop body), `roofline` for the dominant kernel -- the MFMA zgemm of the LU trailing updates, every
K>=256 launch bracketed by HIP events on the stream it runs on inside the timed region and rated over the union
of those intervals, plus an untimed single-stream pass where every kernel is timed alone -- and, at N=1,
`small_batch_rates` (the per-rank shares of 8/4/2-way sharding on one GPU) and a `cpu_baseline` object: the
NumPy/SciPy oracle timed on this host's cores on a bounded sample.
"""
import argparse
import json
import os
import random
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

FP64_MFMA_PEAK_TFLOPS = 78.6     # MI355X dense fp64 matrix peak (AMD spec)
HBM_PEAK_GBS = 8000.0
PMC_ZGEMM = "r04_zgemm_pmc_traffic.json"
PMC_KERNELS = "r04_pmc_traffic_per_kernel.json"
PMC_MFMA = "r04_pmc_mfma_lds_per_kernel.json"


def cpu_baseline(A, n, budget_s=25.0):
    """Oracle (NumPy/SciPy restatement of AMS:145-331, bit-checked against the reference in
    tests/) timed on this host: whole candidate steps, first attempt succeeds, including the two
    rand(N,N) draws, the N x N temporaries and zgecon that the reference performs."""
    from oracle import maus_oracle as orc
    blas = "unknown"
    try:
        from threadpoolctl import threadpool_info
        pools = threadpool_info()
        threads = max([p.get("num_threads", 1) for p in pools] or [1])
        blas = "; ".join(sorted({f"{p.get('internal_api', '?')} {p.get('version', '?')} ({p.get('architecture', '?')})"
                                 for p in pools if p.get("user_api") == "blas"})) or "unknown"
    except Exception:
        threads = os.cpu_count() or 1
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except Exception:
        pass
    import scipy
    st_np, st_py = np.random.get_state(), random.getstate()
    orc.seed_all(4242)
    strat = {"overall_psi_aggression_factor": 1.0, "max_psi_retries": 25, "current_convergence_threshold": 1e-8,
             "convergence_tolerance": 1e-8}
    know = {"local_solver_preference": orc.DIRECT, "is_sparse_problem": False, "is_hermitian": False}
    ncand = 4
    cands = [orc.new_candidate(A, orc.EIGENVALUE, n) for _ in range(ncand)]
    steps = 0
    t0 = time.perf_counter()
    while True:
        for c in cands:
            orc.candidate_step(c, A, None, strat, know)
            steps += 1
        el = time.perf_counter() - t0
        if el > budget_s or steps >= 64:
            break
        if el / steps * (steps + ncand) > budget_s * 1.15:
            break
    el = time.perf_counter() - t0
    np.random.set_state(st_np); random.setstate(st_py)
    return {"value": steps / el, "unit": "candidate-steps/s", "cores": int(threads), "kind": "port",
            "cpu_model": cpu_model, "blas": blas, "numpy": np.__version__, "scipy": scipy.__version__,
            "note": "kind 'port': the NumPy/SciPy oracle (a restatement of the reference checked bit for bit against fixtures captured "
                    "from it), not the reference file itself, which never leaves the build container; a reported baseline, not credit",
            "sample": f"{steps} whole candidate steps ({ncand} candidates x {steps // ncand} iterations) of the NumPy/SciPy oracle at n={n}, "
                      f"{el:.1f} s, BLAS threads={threads}, host cpus={os.cpu_count()}"}


def self_launch(args) -> int:
    """`python bench.py --gpus N` with N > 1 outside torchrun: start N ranks as CHILD processes (one per GPU) and
    return their exit code.  Runs before anything in this process has touched the GPU (nothing is exec'ed)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MAUS_JOB_SECRET", os.urandom(16).hex())        # dist.py: the ranks' RCCL id exchange trusts nobody without it
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------------------------------
# The other BASELINE.json configurations: same JSON schema, the roofline of THAT configuration's
# dominant kernel, the oracle timed on the same workload.  The timed loop bodies start at iteration 1 of a fresh
# population (warm-up runs on a scratch population of the same engine: workspace, eigendecomposition, tuned streams).
# ---------------------------------------------------------------------------------------------------------------------
OTHER = {
    "c2": dict(n=1024, pop=256, steps=10, kind="eig", what="1024x1024 dense non-Hermitian eig (complex128 Ginibre/sqrt(n)), direct-LU path"),
    "c3": dict(n=4096, pop=512, steps=12, kind="lin", what="4096x4096 linear system diag(10^U(0,7) e^{2 pi i U}) + 0.1 Ginibre/sqrt(n) (cond ~1e7, "
                                                          "'Fragile': GMRES preferred), stuck_counter = 2 preset before every step so that the "
                                                          "Jacobi preconditioner is active (AMS:65-72)"),
    "c4": dict(n=8192, pop=128, steps=3, kind="herm", what="8192x8192 Hermitian eig (B+B^H)/2, Hermitian shortcut (AMS:155-181), one GPU's share "
                                                            "(128) of the 1024 candidates BASELINE.json shards 8 ways"),
    "c5": dict(n=2048, pop=512, steps=20, kind="svd", what="2048x2048 complex SVD power step on U diag(logspace(0,-8)) V^H (cond 1e8)"),
}


def _oracle_baseline(kind, A, b, n, budget_s, max_steps, herm_sample_n=None):
    """cpu_baseline of a side configuration: the oracle's candidate_step on the same matrix, bounded.  herm_sample_n: one
    Hermitian candidate step is one eigh of the whole matrix (AMS:161: 73 s at n = 8192 on this host's quota of cores), so the
    default bench line times it on the leading herm_sample_n x herm_sample_n block and scales by (n / herm_sample_n)^3 --
    labelled as an extrapolation; `--config c4` alone times the real thing."""
    from oracle import maus_oracle as orc
    import scipy
    if kind == "herm" and herm_sample_n is not None and herm_sample_n < n:
        sub = np.ascontiguousarray(A[:herm_sample_n, :herm_sample_n])
        r = _oracle_baseline(kind, sub, None, herm_sample_n, budget_s, 1)
        scale = (n / float(herm_sample_n)) ** 3
        r["value"] /= scale
        r["kind"] = "port (extrapolated)"
        r["sample"] = (f"EXTRAPOLATED: {r['sample']} -- on the leading {herm_sample_n} x {herm_sample_n} block, divided by (n / {herm_sample_n})^3 = {scale:.0f} "
                       f"(one step = one O(n^3) eigh, AMS:161)")
        return r
    try:
        from threadpoolctl import threadpool_info
        pools = threadpool_info()
        threads = max([p.get("num_threads", 1) for p in pools] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    st_np, st_py = np.random.get_state(), random.getstate()
    orc.seed_all(4242)
    okind = {"eig": orc.EIGENVALUE, "herm": orc.EIGENVALUE, "lin": orc.SOLVE_LINEAR_SYSTEM, "svd": orc.SVD}[kind]
    strat = {"overall_psi_aggression_factor": 10.0 if kind == "lin" else (2.0 if kind == "svd" else 1.0), "max_psi_retries": 25,
             "current_convergence_threshold": {"lin": 1e-4, "svd": 1e-5}.get(kind, 1e-8), "convergence_tolerance": 1e-8}
    know = {"local_solver_preference": orc.GMRES if kind == "lin" else orc.DIRECT, "is_sparse_problem": False,
            "is_hermitian": kind == "herm"}
    ncand = 4 if kind != "herm" else 1
    cands = [orc.new_candidate(A, okind, n) for _ in range(ncand)]
    steps, t0 = 0, time.perf_counter()
    while True:
        for c in cands:
            if kind == "lin":
                c.stuck = 2
            if kind == "herm":
                c.state = orc.EXPLORING                    # the reference decomposes the matrix again for every stepped candidate (AMS:161)
            orc.candidate_step(c, A, b, strat, know, gmres_mode="rtol")
            steps += 1
        el = time.perf_counter() - t0
        if el > budget_s or steps >= max_steps or el / steps * (steps + ncand) > budget_s * 1.15:
            break
    el = time.perf_counter() - t0
    np.random.set_state(st_np); random.setstate(st_py)
    return {"value": steps / el, "unit": "candidate-steps/s", "cores": int(threads), "kind": "port",
            "numpy": np.__version__, "scipy": scipy.__version__,
            "sample": f"{steps} whole candidate steps of the NumPy/SciPy oracle (a restatement of the reference, checked bit for bit against "
                      f"fixtures captured from it; not the reference itself) on the same matrix, {el:.1f} s, BLAS threads={threads}"}


def run_other_config(args):
    import scenarios
    from adaptive_matrix_solver_amd.solver import MAUS_Solver, ProblemType, SolutionCandidate
    cfg = OTHER[args.config]
    n = cfg["n"] if args.n is None else args.n
    P = cfg["pop"] if args.pop is None else args.pop
    kind = cfg["kind"]
    steps_k = args.steps if "--steps" in sys.argv else cfg["steps"]
    # Hermitian shortcut: every candidate converges in its first step, so a warm-up body on a scratch solver warms nothing
    # that the timed solver meets again -- and a second solver on the same engine would put a host comparison of two copies of
    # the matrix (is the resident decomposition still this matrix's?  ~100 ms at n = 8192) into the first timed loop body
    warm_k = args.warmup if "--warmup" in sys.argv else (0 if kind == "herm" else 1)
    b = None
    if kind == "eig":
        A, PT = scenarios.ginibre(n, n), ProblemType.EIGENVALUE
    elif kind == "lin":
        (A, b), PT = scenarios.wide_diag_system(n, n, decades=7.0, offdiag=0.1), ProblemType.SOLVE_LINEAR_SYSTEM
    elif kind == "herm":
        A, PT = scenarios.hermitian(n, n), ProblemType.EIGENVALUE
    else:
        A, PT = scenarios.prescribed_svd(n, n, n, -8.0), ProblemType.SVD

    def build(engine=None, diag=None):
        np.random.seed(1234); random.seed(1234); SolutionCandidate._candidate_id_counter = 0
        return MAUS_Solver(A, PT, b_vector=b, initial_num_candidates=P, global_convergence_tol=1e-8, quiet=True,
                           engine=engine, diag_info=diag)

    t0 = time.perf_counter()
    scratch = build()
    t_build = time.perf_counter() - t0
    eng, ctx = scratch.engine, scratch.engine.ctx
    info = ctx.device_info()
    S = SolutionCandidate.State

    def body(solver, it):
        if kind == "lin":                        # Jacobi is built only for stuck_counter > 1 (AMS:65); a successful step decrements it
            for c in solver.candidates:
                if c.state not in (S.CONVERGED, S.RETIRED):
                    c.stuck_counter = 2
        return solver.loop_body(it)

    t0 = time.perf_counter()
    for it in range(warm_k):
        body(scratch, it + 1)
    ctx.sync()
    t_warm = time.perf_counter() - t0                  # c4: contains the once-per-matrix eigh unless the diagnostics already did it
    diag = dict(scratch.diag_info)
    if warm_k == 0:
        solver = scratch                               # as a user runs it: one solver, construction outside the metric
    else:
        del scratch
        solver = build(engine=eng, diag=diag)
    gm = {"calls": 0, "cands": 0, "inner": 0}
    if kind == "lin":
        _g = ctx.gmres

        def _gmres(*a, **k):
            info_, inner, status = _g(*a, **k)
            gm["calls"] += 1; gm["cands"] += len(inner); gm["inner"] += int(np.sum(inner))
            return info_, inner, status
        ctx.gmres = _gmres
    ctx.sync()
    per_step, steps_done = [], 0
    t0 = time.perf_counter()
    for it in range(steps_k):
        ts = time.perf_counter()
        act = body(solver, it + 1)
        per_step.append({"ms": round((time.perf_counter() - ts) * 1e3, 3), "active": act})
        steps_done += act
    ctx.sync()
    elapsed = time.perf_counter() - t0
    # one more population from iteration 1 with every launch bracketed by HIP events: per-class kernel times (untimed)
    prof_solver = build(engine=eng, diag=diag)
    ctx.profile_enable(1)
    ctx.sync()
    prof_steps = 0
    # one stream for this pass: where the timed run splits an LU batch over two sub-batch streams (n <= 1024, more matrices than
    # CUs) the event-bracketed durations of kernels that ran side by side would each contain the other's time
    streams_env = os.environ.get("MAUS_LU_STREAMS")
    os.environ["MAUS_LU_STREAMS"] = "1"
    try:
        for it in range(min(steps_k, 3)):
            prof_steps += body(prof_solver, it + 1)
        ctx.sync()
    finally:
        if streams_env is None:
            del os.environ["MAUS_LU_STREAMS"]
        else:
            os.environ["MAUS_LU_STREAMS"] = streams_env
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    kernel_ms = {k: round(v["ms"], 3) for k, v in prof.items() if v["ms"] > 0}
    tot_ms = sum(v["ms"] for v in prof.values())
    # algorithmic flops per candidate step
    if kind == "eig":
        step_flops = 8.0 / 3.0 * n ** 3 + 24.0 * n * n
    elif kind == "lin":
        k_avg = gm["inner"] / max(1, gm["cands"])
        step_flops = 8.0 * n * n * (k_avg + 2.0) + 16.0 * n * (k_avg * (k_avg + 1) / 2.0)
    elif kind == "herm":
        step_flops = 16.0 * n * n
    else:
        step_flops = 32.0 * n * n
    # dominant kernel class of the profiled pass and its roofline
    gemm_classes = [k for k in prof if k.startswith("zgemm")]
    dom = max(prof, key=lambda k: prof[k]["ms"])
    if kind != "eig":
        dom = "zgemm" if prof["zgemm"]["ms"] >= 0.25 * tot_ms else dom
    d = prof[dom]
    if dom in gemm_classes:
        # LU trailing updates and (since round 3) the population products run the 3M kernel: 6 M N K executed per complex GEMM
        use3m = True
        exec_ratio = 0.75
        alg = d["flops"] / max(1e-12, d["ms"] * 1e-3) / 1e12
        what = "3M LU trailing-update zgemm" if kind == "eig" else f"{'3M' if use3m else '4M'} population zgemm: A@X / A^H U / conj(X) V"
        roof = {"bound": "mfma", "kernel": f"{dom} ({what}, v_mfma_f64_16x16x4_f64)",
                "achieved": alg * exec_ratio, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": alg * exec_ratio / FP64_MFMA_PEAK_TFLOPS,
                "achieved_algorithmic_8mnk": alg, "launches": d["launches"], "avg_launch_ms": d["ms"] / max(1, d["launches"]),
                "traffic": None, "kernel_time_share": d["ms"] / tot_ms if tot_ms > 0 else None,
                "algorithmic_bytes_per_launch": d["bytes"] / max(1, d["launches"]),
                "hbm_GBs_at_algorithmic_bytes": d["bytes"] / max(1e-12, d["ms"] * 1e-3) / 1e9}
    else:
        gbs = d["bytes"] / max(1e-12, d["ms"] * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": dom, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                "launches": d["launches"], "avg_launch_ms": d["ms"] / max(1, d["launches"]), "traffic": None,
                "kernel_time_share": d["ms"] / tot_ms if tot_ms > 0 else None,
                "algorithmic_bytes_per_launch": d["bytes"] / max(1, d["launches"])}
    roof["source"] = (f"one untimed single-stream pass of {min(steps_k, 3)} loop bodies ({prof_steps} candidate steps) from iteration 1 with every launch "
                      "bracketed by HIP events on the context's stream")
    out = {
        "metric": f"candidate-steps/sec, BASELINE.json configs[{int(args.config[1]) - 1}] ({args.config})",
        "value": steps_done / elapsed, "unit": "candidate-steps/s", "n_gpus": 1, "steps": steps_k, "warmup": warm_k,
        "ms_per_step": elapsed / max(1, steps_k) * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{cfg['what']}, initial_num_candidates={P}", "n": n, "pop": P, "candidate_steps_timed": steps_done,
                   "device": info["name"], "solver_build_s": round(t_build, 2), "warmup_s": round(t_warm, 2),
                   "preferred_solver": solver.problem_knowledge.get("local_solver_preference"),
                   "condition_number": float(solver.cond_number)},
        "per_step": per_step,
        "roofline": roof,
        "kernel_ms_profiled_pass": kernel_ms,
        "step_flops_algorithmic": step_flops,
        "step_tflops": step_flops * steps_done / elapsed / 1e12,
        "step_frac_of_mfma_peak": step_flops * steps_done / elapsed / 1e12 / FP64_MFMA_PEAK_TFLOPS,
        # what reaches the matrix pipe: 3M products (0.75), and for the SVD step two of the reference's four products per
        # candidate step -- the power step takes A v from the previous residual and the residual A^H u from the power step
        # (the same products of the same vectors, bit for bit: csrc/capi.hip av_* / ahu_*)
        "step_frac_of_mfma_peak_executed": (0.75 * (0.5 if kind == "svd" else 1.0)) * step_flops * steps_done / elapsed / 1e12 / FP64_MFMA_PEAK_TFLOPS,
        **({"step_note": "algorithmic flops count the reference's four products per SVD step (AMS:228, 240, 295, 298); two are executed"}
           if kind == "svd" else {}),
    }
    if kind == "lin":
        out["gmres"] = {"calls": gm["calls"], "candidate_solves": gm["cands"], "inner_iterations_mean": gm["inner"] / max(1, gm["cands"])}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = _oracle_baseline(kind, A, b, n, args.cpu_budget, 64, herm_sample_n=2048 if args.side else None)
    print(json.dumps(out))


SIDE_ORDER = ("c2", "c3", "c5", "c4")


def run_side_configs(args):
    """Short runs of the other BASELINE.json configurations, one child process each (`bench.py --config cX --side`: its own
    context, nothing shared with the headline run), condensed into one entry per configuration.  c5 runs the population the
    reference itself would build (AMS:366: at least 3 min(rows, cols) = 6 144 candidates, not BASELINE's 512); c4 one GPU's
    share (128) of its 1 024 candidates; c3's first loop body (twice as long as the later ones) is inside its timed region."""
    res = {}
    t_all = time.perf_counter()
    for name in SIDE_ORDER:
        cmd = [sys.executable, os.path.abspath(__file__), "--config", name, "--side", "--cpu-budget", "4"]
        if args.no_cpu_baseline:
            cmd.append("--no-cpu-baseline")
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
            line = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not line:
                res[name] = {"error": f"exit code {p.returncode}", "stderr_tail": p.stderr.decode(errors="replace")[-400:]}
                continue
            d = json.loads(line[-1])
        except Exception as e:                                   # noqa: BLE001 -- a side run must not take the headline line down
            res[name] = {"error": f"{type(e).__name__}: {e}"}
            continue
        r = d["roofline"]
        ent = {"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
               "steps": d["steps"], "warmup": d["warmup"], "candidate_steps_timed": d["config"]["candidate_steps_timed"],
               "per_step_ms": [x["ms"] for x in d["per_step"]], "per_step_active": [x["active"] for x in d["per_step"]],
               "roofline": {"bound": r["bound"], "kernel": r["kernel"], "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"],
                            "frac": r["frac"], "kernel_time_share": r.get("kernel_time_share")},
               "step_frac_of_mfma_peak": d["step_frac_of_mfma_peak"], "step_frac_of_mfma_peak_executed": d.get("step_frac_of_mfma_peak_executed"),
               **({"step_note": d["step_note"]} if "step_note" in d else {}), "kernel_ms_profiled_pass": d["kernel_ms_profiled_pass"],
               "solver_build_s": d["config"]["solver_build_s"], "wall_s_of_this_side_run": round(time.perf_counter() - t0, 1)}
        if "gmres" in d:
            ent["gmres"] = d["gmres"]
        if "cpu_baseline" in d:
            cb = d["cpu_baseline"]
            ent["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample"]}
        res[name] = ent
    res["wall_s_total"] = round(time.perf_counter() - t_all, 1)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", dest="n", type=int, default=None, help="matrix order (default: the configuration's; the metric's 4096 for c1)")
    ap.add_argument("--pop", type=int, default=None, help="initial_num_candidates (default: the configuration's; the metric's 256 for c1)")
    ap.add_argument("--config", choices=["c1", "c2", "c3", "c4", "c5"], default="c1",
                    help="c1 (default): the metric's configuration, n=4096 dense eig, pop=256.  The other BASELINE.json "
                         "configurations, same JSON schema, single GPU: c2 = 1024x1024 eig / 256 candidates (batched LU); "
                         "c3 = 4096x4096 linear system / 512 candidates, GMRES + Jacobi; c4 = 8192x8192 Hermitian eig (the "
                         "shortcut; --pop defaults to one GPU's share of 1024, 128); c5 = 2048x2048 SVD cond 1e8 / 512 candidates")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-history", action="store_true",
                    help="record_history=False: keep only residual_history (the reference appends every iterate to param_history, "
                         "AMS:303; the default here does too, into the device-backed store)")
    ap.add_argument("--no-isolated", action="store_true", help="skip the untimed single-stream kernel-timing pass")
    ap.add_argument("--no-small-batch", action="store_true", help="skip the pop = 32/64/128 side runs (N=1 only)")
    ap.add_argument("--kernel-events", choices=["sampled", "all", "off"], default="sampled",
                    help="HIP-event bracketing of kernel launches in the timed region: every K>=256 zgemm launch (default), "
                         "every launch of every kernel (costs 3-5 %% of throughput), or none")
    ap.add_argument("--cpu-budget", type=float, default=40.0)
    ap.add_argument("--no-side", action="store_true",
                    help="default config only: skip the short runs of the other BASELINE.json configurations (c2, c3, c5, c4) that "
                         "are folded into the JSON line as `side_configs` (N=1)")
    ap.add_argument("--side", action="store_true", help="(internal) this process is one of those short runs: bounded CPU baseline")
    ap.add_argument("--skip-diagnosis", action="store_true",
                    help="profiling aid: construct the solver with the known start-up diagnostics of the metric's matrix "
                         "(dense, non-Hermitian, 'Stable') instead of running the condition estimator, whose ten "
                         "single-matrix LUs would otherwise be mixed into rocprofv3's per-kernel averages; the timed loop "
                         "bodies are the same")
    ap.add_argument("--launch-check", action="store_true",
                    help="start the ranks, form the communicator, all-gather the rank ids, print {n_gpus} and exit "
                         "(with MAUS_DIST_BACKEND=gloo without touching a GPU: tests/test_dist_gloo.py checks the --gpus N "
                         "self-launch with it)")
    args = ap.parse_args()

    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))            # children print the JSON line; nothing here has touched the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; start one rank per GPU "
                 f"(python bench.py --gpus N does that itself)")
    comm = None
    # 'rccl' (alias 'nccl'): the library's own RCCL collectives, one rank per GPU, no torch in the process;
    # MAUS_DIST_BACKEND=gloo lets several ranks share one GPU (or none: --launch-check) for rehearsal
    backend = {"nccl": "rccl"}.get(os.environ.get("MAUS_DIST_BACKEND", "rccl"), os.environ.get("MAUS_DIST_BACKEND", "rccl"))
    if args.launch_check:
        ranks = [0]
        if world > 1:
            from adaptive_matrix_solver_amd import dist as mdist
            comm = mdist.init_from_env(backend)
            if comm.transport == "rccl":          # the communicator lives on a device context (one per GPU)
                from adaptive_matrix_solver_amd import Context
                comm.attach(Context(local_rank))
            ranks = comm.allgather_rows(np.array([[float(rank)]]), [1] * world)[:, 0].astype(int).tolist()
            comm.barrier()
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "ranks": ranks, "backend": backend if world > 1 else None}))
        return
    if world > 1:
        from adaptive_matrix_solver_amd import dist as mdist
        comm = mdist.init_from_env(backend)

    import scenarios
    from adaptive_matrix_solver_amd.solver import MAUS_Solver, ProblemType, SolutionCandidate

    if args.config != "c1":
        if world != 1:
            sys.exit("bench.py: --config c2..c5 are single-GPU side benches (the driver's scaling run is the default config)")
        run_other_config(args)
        return
    args.n = 4096 if args.n is None else args.n
    args.pop = 256 if args.pop is None else args.pop
    n, P = args.n, args.pop
    A = scenarios.ginibre(n, n)                 # (G1 + i G2)/sqrt(n), seed n  
    device = local_rank if backend == "rccl" else 0

    def build(pop, engine=None, diag=None):
        np.random.seed(1234); random.seed(1234); SolutionCandidate._candidate_id_counter = 0
        return MAUS_Solver(A, ProblemType.EIGENVALUE, initial_num_candidates=pop, global_convergence_tol=1e-8,
                           device=device, pert_mode="auto", comm=comm, quiet=True, record_history=not args.no_history,
                           engine=engine, diag_info=diag)

    t_build = time.perf_counter()
    diag0 = None
    if args.skip_diagnosis:
        diag0 = {"is_hermitian": False, "is_complex_symmetric": False, "is_sparse_init": False, "condition_number": 1.0e4,
                 "is_singular": False, "condition_number_is_estimate": True, "note": "--skip-diagnosis placeholder"}
    solver = build(P, diag=diag0)
    t_build = time.perf_counter() - t_build
    ctx = solver.engine.ctx
    info = ctx.device_info()

    def sync_all():
        ctx.sync()                               # every kernel and collective of this rank runs on the context's stream
        if comm is not None:
            comm.barrier()

    it = 0
    for _ in range(args.warmup):
        it += 1
        solver.loop_body(it)
    if args.kernel_events == "sampled":
        # every K>=256 zgemm launch is bracketed (a few dozen per loop body on any rank): the union of the bracketed
        # intervals is only meaningful when none is skipped
        os.environ["MAUS_PROF_STRIDE"] = "1,0"
    mode = {"sampled": 2, "all": 1, "off": 0}[args.kernel_events]
    # wall time inside maus_shifted_lu_solve per loop body (the rest of a loop body: Rayleigh / relax / residual phases,
    # the host's RNG-event replay and population bookkeeping, collectives)
    lu_wall = [0.0]
    _lu = ctx.shifted_lu_solve

    def _timed_lu(*a, **k):
        t = time.perf_counter()
        try:
            return _lu(*a, **k)
        finally:
            lu_wall[0] += time.perf_counter() - t
    ctx.shifted_lu_solve = _timed_lu
    ctx.profile_enable(mode)
    sync_all()
    comm0 = comm.stats() if comm is not None else None
    t0 = time.perf_counter()
    steps_done = 0
    per_step = []
    cum_union = cum_flops = 0.0
    for _ in range(args.steps):
        it += 1
        ts = time.perf_counter()
        lu_wall[0] = 0.0
        c_before = comm.stats() if comm is not None else None
        act = solver.loop_body(it)               # ends synchronously: the host fetched every phase's results
        te = time.perf_counter()
        steps_done += act
        rec = {"ms": round((te - ts) * 1e3, 3), "active": act, "lu_call_ms": round(lu_wall[0] * 1e3, 3)}
        if comm is not None:
            c_after = comm.stats()
            rec["collective_ms"] = round(c_after["ms"] - c_before["ms"], 3)
            rec["collectives"] = c_after["collectives"] - c_before["collectives"]
        if mode:
            z = ctx.profile_read_class(0)        # cumulative since profile_enable
            rec["k256_union_ms"] = round(z["union_ms"] - cum_union, 3)
            rec["k256_tflops"] = round((z["flops"] - cum_flops) / max(1e-9, (z["union_ms"] - cum_union) * 1e-3) / 1e12, 2)
            cum_union, cum_flops = z["union_ms"], z["flops"]
        per_step.append(rec)
    sync_all()
    elapsed = time.perf_counter() - t0
    per_rank = None
    if comm is not None:
        # max over ranks of the timed region; and every rank's own view of it (loop-body wall times, time inside
        # maus_shifted_lu_solve, time inside collectives) gathered to rank 0 for the JSON line
        c1 = comm.stats()
        mine = np.array([[elapsed, (c1["ms"] - comm0["ms"]) * 1e-3, float(c1["collectives"] - comm0["collectives"]),
                          float(c1["bytes"] - comm0["bytes"])] +
                         [r["ms"] for r in per_step] + [r["lu_call_ms"] for r in per_step] + [r["collective_ms"] for r in per_step]])
        allr = comm.allgather_rows(mine, [1] * world)
        elapsed = float(allr[:, 0].max())
        K = args.steps
        per_rank = [{"rank": r, "elapsed_s": round(float(allr[r, 0]), 4), "collective_s": round(float(allr[r, 1]), 4),
                     "collectives": int(allr[r, 2]), "collective_bytes": int(allr[r, 3]),
                     "loop_body_ms": [round(float(x), 2) for x in allr[r, 4:4 + K]],
                     "lu_call_ms": [round(float(x), 2) for x in allr[r, 4 + K:4 + 2 * K]],
                     "collective_ms": [round(float(x), 3) for x in allr[r, 4 + 2 * K:4 + 3 * K]]} for r in range(world)]
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    ws_allocs = ctx.lu_workspace_allocations()
    # Isolated pass (untimed, rank 0 at N=1 only): one more step on a single stream with every launch bracketed, so
    # that each kernel has the GPU to itself -- in the timed region two sub-batch streams overlap and a kernel's
    # event-to-event time includes whatever the other stream ran beside it.
    iso, iso_active = None, 0
    if world == 1 and args.kernel_events != "off" and not args.no_isolated:
        saved = os.environ.get("MAUS_LU_STREAMS")
        os.environ["MAUS_LU_STREAMS"] = "1"
        ctx.profile_enable(1)
        ctx.sync()
        iso_active = solver.loop_body(it + 1)
        ctx.sync()
        iso = ctx.profile_read()
        ctx.profile_enable(False)
        if saved is None:
            os.environ.pop("MAUS_LU_STREAMS", None)
        else:
            os.environ["MAUS_LU_STREAMS"] = saved

    # Small per-rank populations on this one GPU (what each rank of an 8 / 4 / 2-way sharded run executes per step):
    # the ceiling of strong scaling before any communication.  Same matrix, same engine (workspace already sized).
    small = None
    if world == 1 and not args.no_small_batch and n == 4096 and P == 256:
        small = {}
        for sp in (32, 64, 128):
            s2 = build(sp, engine=solver.engine, diag=solver.diag_info)
            s2.loop_body(1)
            ctx.sync()
            ts = time.perf_counter()
            cs = sum(s2.loop_body(2 + k) for k in range(3))
            ctx.sync()
            small[str(sp)] = round(cs / (time.perf_counter() - ts), 2)
            del s2

    if rank == 0:
        # Dominant kernel: the zgemm launches of the LU trailing updates proper (profile class "zgemm": K >= 256,
        # ~70 % of the step).  The small-K launches of the panel recursion (classes zgemm_k128..k16, bandwidth-bound)
        # are the same kernel template; they are timed in the isolated pass, where nothing overlaps them.
        gk = [k for k in prof if k.startswith("zgemm")]
        g = dict(prof["zgemm"])
        tot_ms = sum(v["ms"] for v in prof.values())
        exec_ratio = 0.75       # real flops executed on the matrix pipe / algorithmic 8MNK (3M complex products)
        # rate = flops of the bracketed K>=256 launches / time during which at least one of them was executing (the
        # union of their intervals over both sub-batch streams).  Two trailing updates running side by side share the
        # machine: the plain sum of their event-to-event durations would count that time twice, and a sample's duration
        # would depend on what the other stream happened to run beside it.
        busy_ms = g.get("union_ms", 0.0) or g["ms"]
        alg = (g["flops"] / (busy_ms * 1e-3) / 1e12) if busy_ms > 0 else 0.0
        alg_sum = (g["flops"] / (g["ms"] * 1e-3) / 1e12) if g["ms"] > 0 else 0.0
        per_launch_ms = g["ms"] / max(1, g["launches"])
        # HBM bytes per K>=256 zgemm launch from the committed PMC passes (rocprofv3 --pmc cannot be combined with
        # the timed run; profiles/<PMC_ZGEMM> holds the recipe).  Those passes ran the step as ONE chunk on one
        # stream; the timed loop bodies run other batch sizes, so the measured bytes are scaled by the ratio of
        # algorithmic bytes per launch (the traffic / algorithmic ratio is what carries over).
        traffic = None
        traffic_note = None
        for name in (PMC_ZGEMM, "r02_zgemm_pmc_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    pm = json.load(f)
                if n == 4096 and P == 256 and world == 1 and g["launches"] > 0:
                    ratio = pm["hbm_bytes_per_launch"] / pm["algorithmic_bytes_per_launch"]
                    traffic = ratio * g["bytes"] / g["launches"]
                    traffic_note = {"pmc_hbm_bytes_per_launch": pm["hbm_bytes_per_launch"],
                                    "pmc_algorithmic_bytes_per_launch": pm["algorithmic_bytes_per_launch"],
                                    "hbm_over_algorithmic": ratio, "source": "profiles/" + name}
                break
            except Exception:
                continue
        # HBM-bound kernels: PMC bytes per sweep (a sweep of `matrices` solves, recorded in the file) scaled to the
        # number of solves of the isolated pass, over their time in that pass
        hbm_kernels = None
        try:
            if iso is not None and n == 4096 and iso_active > 0:
                with open(os.path.join(ROOT, "profiles", PMC_KERNELS)) as f:
                    pk = json.load(f)
                scale = iso_active / float(pk["matrices"])
                hbm_kernels = {"matrices_in_isolated_pass": iso_active, "pmc_matrices": pk["matrices"],
                               "source": "profiles/" + PMC_KERNELS}
                for name in ("laswp", "trsm", "build_h", "lu_panel", "backsolve"):
                    if name in pk["classes"] and iso.get(name, {}).get("ms", 0) > 0:
                        by = pk["classes"][name]["hbm_bytes"] * scale
                        gbs = by / (iso[name]["ms"] * 1e-3) / 1e9
                        hbm_kernels[name] = {"ms_per_sweep": round(iso[name]["ms"], 3), "hbm_bytes_per_sweep": by,
                                             "achieved_GBs": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 3)}
        except Exception:
            hbm_kernels = None
        pmc_mfma = None
        try:
            with open(os.path.join(ROOT, "profiles", PMC_MFMA)) as f:
                pj = json.load(f)
                pm = pj[[k for k in pj if k.startswith("zgemm3m_dma_kernel<2, 3, 2, 2, true")][0]]
            pmc_mfma = {"kernel": "zgemm3m_dma_kernel<2, 3, 2, 2, true> (64x64 tiles: the K = 512 outer updates with M, N >= 1536)",
                        "MfmaUtil": round(pm["mfma_util"], 4), "effective_clock_GHz": round(pm["effective_clock_GHz"], 3),
                        "mfma_tflops_executed": round(pm["mfma_tflops_executed"], 2),
                        "mfma_peak_tflops_at_effective_clock": round(pm["mfma_peak_tflops_at_effective_clock"], 2),
                        "lds_bank_conflict_per_idx_active": pm.get("lds_bank_conflict_per_idx_active"),
                        "wave_wait_any_share": round(pm["wait_any_share"], 3), "wave_wait_inst_share": round(pm["wait_inst_any_share"], 3),
                        "source": "profiles/" + PMC_MFMA + " (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs); 256 solves, profiled run)"}
        except Exception:
            pmc_mfma = None
        ms_norm = [r["ms"] / max(1, r["active"]) for r in per_step]
        out = {
            "metric": "candidate-steps/sec, n=4096 dense eig pop=256, 1/2/4/8 GPUs vs CPU ref",
            "value": steps_done / elapsed, "unit": "candidate-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / max(1, args.steps) * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"n={n} dense non-Hermitian eig (complex128 Ginibre/sqrt(n)), initial_num_candidates={P}, "
                                   "direct-LU InverseIterateSolver path", "n": n, "pop": P,
                       "candidate_steps_timed": steps_done,
                       "parallelism": "A replicated, active candidates block-sharded over ranks, all-gather of records per phase",
                       "pert_mode": ("mt19937 (the reference's 2 x rand(N,N) draws per attempt regenerated bit-identically on the "
                                     "device from the NumPy state)") if n > 256 else "uniform (host draws uploaded)",
                       "device": info["name"], "solver_build_s": round(t_build, 2)},
            "per_step": per_step,
            "per_rank": per_rank,
            "per_step_summary": {"ms_per_candidate_step_median": round(float(np.median(ms_norm)), 4),
                                 "ms_per_candidate_step_max": round(float(np.max(ms_norm)), 4),
                                 "lu_workspace_allocations_total": ws_allocs},
            "roofline": {"bound": "mfma", "kernel": "zgemm3m_dma_kernel (LDS-DMA staged 3M zgemm, 64x64 / 64x32 tiles), K>=256 launches (LU trailing updates, v_mfma_f64_16x16x4_f64)",
                         # achieved / frac: real flops EXECUTED on the matrix pipe (3M: 6*M*N*K per complex GEMM) over the
                         # union of the launches' intervals; the algorithmic (8*M*N*K) rate is a side field
                         "achieved": alg * exec_ratio, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": alg * exec_ratio / FP64_MFMA_PEAK_TFLOPS,
                         "achieved_algorithmic_8mnk": alg, "frac_algorithmic_8mnk": alg / FP64_MFMA_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_detail": traffic_note,
                         "algorithmic_bytes_per_launch": g["bytes"] / max(1, g["launches"]),
                         "launches": g["launches"], "avg_launch_ms": per_launch_ms, "event_sampling": args.kernel_events,
                         "busy_ms_union_over_streams": busy_ms, "sum_of_launch_ms": g["ms"],
                         "achieved_algorithmic_by_sum_of_launch_durations": alg_sum,
                         "flops_per_launch_algorithmic": g["flops"] / max(1, g["launches"]),
                         "kernel_time_share": ((g["ms"] / tot_ms) if tot_ms > 0 else None) if args.kernel_events == "all" else None,
                         "measured_mfma_f64_issue_rate_tflops": 77.9,
                         # hardware counters of the K = 512 outer updates (separate rocprofv3 --pmc passes, committed)
                         "pmc_counters": pmc_mfma,
                         "lu_streams": os.environ.get("MAUS_LU_STREAMS", "1"),
                         "isolated_single_stream_pass": None if iso is None else {
                             "matrices": iso_active,
                             "achieved_algorithmic_8mnk": iso["zgemm"]["flops"] / max(1e-9, iso["zgemm"]["ms"] * 1e-3) / 1e12,
                             "avg_launch_ms": iso["zgemm"]["ms"] / max(1, iso["zgemm"]["launches"]),
                             "launches": iso["zgemm"]["launches"],
                             "achieved_algorithmic_all_zgemm_launches": sum(iso[k]["flops"] for k in gk) / max(1e-9, sum(iso[k]["ms"] for k in gk) * 1e-3) / 1e12,
                             "all_zgemm_launches": sum(iso[k]["launches"] for k in gk),
                             "kernel_ms": {k: round(v["ms"], 3) for k, v in iso.items()},
                             "note": "one extra untimed step, MAUS_LU_STREAMS=1, every launch bracketed by HIP events"},
                         "flop_convention": ("achieved = real flops executed on the matrix pipe: the kernel forms each complex product "
                                             "from 3 real MFMA products (3M), 6*M*N*K per complex GEMM; *_algorithmic_8mnk counts the "
                                             "8*M*N*K of the textbook complex product")},
            ("kernel_ms" if args.kernel_events == "all" else "kernel_ms_estimated_from_sampled_launches"): {k: round(v["ms"], 3) for k, v in prof.items()},
            "hbm_bound_kernels": hbm_kernels,
            "step_tflops": (8.0 / 3.0 * n ** 3 + 24.0 * n * n) * steps_done / elapsed / 1e12,
            "step_frac_of_mfma_peak": (8.0 / 3.0 * n ** 3 + 24.0 * n * n) * steps_done / elapsed / 1e12 / FP64_MFMA_PEAK_TFLOPS,
            # the same in the convention of roofline.frac: flops EXECUTED on the matrix pipe (the LU's 8/3 n^3 run as 3M
            # products, 0.75 of them; the 24 n^2 of the population products as 4M)
            "step_frac_of_mfma_peak_executed": (0.75 * 8.0 / 3.0 * n ** 3 + 24.0 * n * n)
                                               * steps_done / elapsed / 1e12 / FP64_MFMA_PEAK_TFLOPS,
            "small_batch_rates": small,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(A, n, args.cpu_budget)
        if world == 1 and not args.no_side and n == 4096 and P == 256:
            # the headline context goes first: its LU workspace holds most of the device memory
            del solver
            ctx.close()
            out["side_configs"] = run_side_configs(args)
        print(json.dumps(out))
    if comm is not None:
        comm.barrier()


if __name__ == "__main__":
    main()
