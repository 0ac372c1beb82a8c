"""Oracle against itself with one ulp added to every matrix it factorises (tests/rounding.py): what a backward-error-sized
difference on the REFERENCE side alone does to the long trajectories whose GPU comparison was loosened in round 2
(VERDICT r02, item 3).  CPU only; single-threaded BLAS so that the run is reproducible.

    python tools/rounding_sensitivity.py > profiles/r03_rounding_sensitivity.txt      (~4 minutes)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
from threadpoolctl import threadpool_limits  # noqa: E402

import rounding  # noqa: E402
import scenarios  # noqa: E402
from oracle import maus_oracle as orc  # noqa: E402

SCEN = {
    "lap8_p96": dict(kind="eig", build=("laplace", 8, 8, False), P=96, iters=60, seed=7, tol=1e-7),
    "lap8": scenarios.TRAJECTORIES["lap8"],
    "eig288_p64": dict(kind="eig", build=("ginibre", 288, 300, None), P=64, iters=16, seed=11, tol=1e-8),
}
SEEDS = {"lap8_p96": (1, 2, 3), "lap8": (1, 2, 3), "eig288_p64": (1, 2, 3, 4)}

with threadpool_limits(limits=1):
    for name, spec in SCEN.items():
        scenarios.TRAJECTORIES[name] = spec
        t0 = time.time()
        ref, anorm = rounding.oracle_run(name, spec["iters"])
        print(f"## {name}: n = {spec['build'][1]}, P = {spec['P']}, {spec['iters']} iterations (reference run {time.time() - t0:.0f} s)")
        print("# per perturbation seed: first iteration with |dlambda| > 1e-9 among stepped candidates; first iteration whose "
              "survivor order differs; first iteration whose integer bookkeeping differs; RNG streams equal throughout")
        rows = []
        for seed in SEEDS[name]:
            got, _ = rounding.oracle_run(name, spec["iters"], perturb_seed=seed, ulps=1)
            D = rounding.drift(ref, got)
            d = np.array([x[0] for x in D])
            f_tol = next((i for i, x in enumerate(d) if x > 1e-9), None)
            f_ord = next((i for i, x in enumerate(D) if not x[2]), None)
            f_int = next((i for i, x in enumerate(D) if not x[1]), None)
            print(f"seed {seed}: drift>1e-9 at {f_tol}; order differs at {f_ord}; bookkeeping differs at {f_int}; rng equal {all(x[3] for x in D)}")
            rows.append(d)
            if f_ord is not None and name == "eig288_p64":
                r, g = ref[f_ord], got[f_ord]
                rb = {x["id"]: x for x in r["rows"]}
                gb = {x["id"]: x for x in g["rows"]}
                only_r, only_g = sorted(set(r["after"]) - set(g["after"])), sorted(set(g["after"]) - set(r["after"]))
                print(f"   iteration {f_ord}: survivors only in the reference run {only_r}, only in the perturbed run {only_g}")
                for i in only_r + only_g:
                    x, y = rb.get(i), gb.get(i)
                    print(f"   id {i}: reference run (state {x['state']}, w {x['w']}, resid {x['resid']:.3e}, lambda {x['lam']:.9f}) | "
                          f"perturbed run (state {y['state']}, w {y['w']}, resid {y['resid']:.3e})")
        env = np.max(rows, axis=0)
        print("# envelope over the seeds, max |dlambda|/max(1,|lambda|) per iteration:")
        print(" ".join(f"{i}:{x:.1e}" for i, x in enumerate(env)))
        lo, hi = (5, min(30, len(env) - 1))
        if env[lo] > 0 and hi > lo:
            print(f"# growth per iteration between {lo} and {hi}: {(env[hi] / env[lo]) ** (1.0 / (hi - lo)):.2f}x")
        print()
