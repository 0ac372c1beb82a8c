#!/usr/bin/env python3
"""Generate tests/golden/*.json|npz by importing the REFERENCE unmodified.

Runs only in the build container (needs /root/reference); the outputs are data
(inputs + expected outputs), committed; the reference's text never enters the
repo.  Harness obligations follow SURVEY §8c:
  * the reference file name is not an identifier -> import by path, no bytecode;
  * zero SolutionCandidate._candidate_id_counter before each capture (F12);
  * seed BOTH np.random and random;
  * drive the loop body (AMS:573-577) by hand -- evolve() itself dies with a
    NameError at AMS:583 (F1);
  * GMRES fixtures only: rebind the module's `spla` to a shim that forwards the
    removed `tol=` keyword as `rtol=` (F2); recorded in the fixture metadata.

Usage:  python tests/golden/make_goldens.py
"""
import importlib.util
import json
import os
import random
import sys
import types

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import scenarios  # noqa: E402
import snapshot  # noqa: E402

REF = "/root/reference/Adaptive_Matrix_Solver_0.1.py"


def load_reference():
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("maus_ref", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def gmres_shim(mod):
    real = mod.spla

    def gmres(A, b, x0=None, tol=1e-5, maxiter=None, M=None):
        return real.gmres(A, b, x0=x0, rtol=tol, maxiter=maxiter, M=M)

    ns = types.SimpleNamespace(gmres=gmres, spsolve=real.spsolve, eigsh=real.eigsh, svds=real.svds,
                               ArpackNoConvergence=real.ArpackNoConvergence)
    return real, ns


def versions():
    return {"numpy": np.__version__, "scipy": scipy.__version__, "python": sys.version.split()[0]}


def seed_all(mod, seed):
    np.random.seed(seed)
    random.seed(seed)
    mod.SolutionCandidate._candidate_id_counter = 0


def rows_of(mod, cands, kind):
    rows = []
    for c in cands:
        if kind == "eig":
            lam, vecs = c.lambda_k, [c.v_k]
        elif kind == "lin":
            lam, vecs = 0j, [c.x_k]
        else:
            lam, vecs = c.sigma_k, [c.u_k, c.right_v_k]
        rows.append({"id": c.id, "state": c.state.value, "stuck": c.stuck_counter,
                     "retries": c.local_psi_retries_needed, "resets": c.num_resets,
                     "w": c.w_k, "resid": c.residual_k, "alpha": c.alpha_local_step,
                     "lam": lam, "vecs": vecs})
    return rows


def ptype(mod, kind):
    return {"eig": mod.ProblemType.EIGENVALUE, "lin": mod.ProblemType.SOLVE_LINEAR_SYSTEM,
            "svd": mod.ProblemType.SVD}[kind]


def run_trajectory(mod, name, spec):
    A, b = scenarios.build(spec)
    seed_all(mod, spec["seed"])
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        solver = mod.MAUS_Solver(A, ptype(mod, spec["kind"]), b_vector=b,
                                 initial_num_candidates=spec["P"], global_convergence_tol=spec["tol"])
    S = mod.SolutionCandidate.State
    out = {"name": name, "spec": {k: v for k, v in spec.items()}, "versions": versions(),
           "gmres_shim": False, "cond": float(solver.cond_number).hex(),
           "hermitian": bool(solver.problem_knowledge["is_hermitian"]),
           "init": {"digest": snapshot.digest_rows(rows_of(mod, solver.candidates, spec["kind"])),
                    "rng": snapshot.rng_digest(),
                    "globals": snapshot.globals_record(solver.landscape_energy, solver.avg_residual,
                                                       solver.avg_stuckness, 0,
                                                       solver.problem_knowledge["numerical_stability_state"],
                                                       solver.problem_knowledge["local_solver_preference"],
                                                       solver.strat_params)},
           "iters": []}
    total_steps = 0
    for it in range(spec["iters"]):
        with contextlib.redirect_stdout(io.StringIO()):
            solver._update_global_diagnostics(it + 1)
            solver._adjust_global_strategy(it + 1)
            steps = 0
            for c in solver.candidates:
                if c.state not in (S.CONVERGED, S.RETIRED):
                    c.update_solution_step(solver.M, solver.b, solver.strat_params, solver.problem_knowledge)
                    steps += 1
            stepped_rows = rows_of(mod, solver.candidates, spec["kind"])
            solver._manage_candidates(it + 1)
        total_steps += steps
        rows = rows_of(mod, solver.candidates, spec["kind"])
        rec = {"steps": steps, "n_after": len(rows),
               "digest_stepped": snapshot.digest_rows(stepped_rows),
               "digest": snapshot.digest_rows(rows), "rng": snapshot.rng_digest(),
               "globals": snapshot.globals_record(solver.landscape_energy, solver.avg_residual,
                                                  solver.avg_stuckness, solver.num_distinct_converged_solutions,
                                                  solver.problem_knowledge["numerical_stability_state"],
                                                  solver.problem_knowledge["local_solver_preference"],
                                                  solver.strat_params),
               "next_id": int(mod.SolutionCandidate._candidate_id_counter)}
        if it < 3 or it == spec["iters"] - 1:
            rec["rows"] = snapshot.full_rows(stepped_rows)
        out["iters"].append(rec)
    out["total_steps"] = total_steps
    return out


def solve_fixtures(mod):
    """G1: InverseIterateSolver.solve, direct path; G5: GMRES(+Jacobi) through the shim.
    Inputs come from scenarios.solve_case_inputs(key); only outputs are stored."""
    arrays = {}
    meta = []
    base = mod.GLOBAL_DEFAULT_PSI_EPSILON_BASE

    def run(key, n, stuck, seed, pref, aggr, shim):
        tgt, rhs = scenarios.solve_case_inputs(key)
        seed_all(mod, seed)
        pos0 = int(np.random.get_state()[2])
        s = mod.InverseIterateSolver(n, base * aggr, 25, pref, False)
        x, att = s.solve(tgt, rhs, stuck)
        arrays[key + "_x"] = x
        meta.append({"key": key, "n": n, "stuck": stuck, "seed": seed, "attempts": int(att), "pref": pref,
                     "aggr": aggr, "max_attempts": 25, "gmres_shim": shim, "rng": snapshot.rng_digest(),
                     "mt_pos": [pos0, int(np.random.get_state()[2])]})

    for n in (8, 16, 64):
        for stuck in (0, 3):
            run(f"direct_n{n}_s{stuck}", n, stuck, 31 + n + stuck, "direct_solve", 1.0, False)
    # large psi: the random perturbation is numerically visible
    run("bigpsi", 16, 8, 4242, "direct_solve", 1e12, False)
    real, shim = gmres_shim(mod)
    mod.spla = shim
    try:
        for n in (32, 128):
            for stuck in (0, 2):
                run(f"gmres_n{n}_s{stuck}", n, stuck, 77 + n + stuck, "iterative_gmres", 10.0, True)
        # hard spectrum, no Jacobi (stuck<=1): GMRES exhausts 50x20 inner iterations -> falls back to LU
        run("gmresfb_n32_s0", 32, 0, 123, "iterative_gmres", 10.0, True)
        # escalated psi (the GMRES analogue of bigpsi): base psi 1e-5, stuck 2 -> psi = 4.6e-5; the random term
        # 0.15*psi*((U1-.5)+i(U2-.5)) of AMS:49-50 is part of H_solve in the GMRES branch too (AMS:52, 89) and is
        # visible in the iterate at the 1e-6 level
        for n in (32, 128):
            run(f"gmresbig_n{n}_s2", n, 2, 555 + n, "iterative_gmres", 1e15, True)
        run("gmresbig_n32_s0", 32, 0, 556, "iterative_gmres", 1e15, True)
    finally:
        mod.spla = real
    # unshimmed: GMRES preferred -> TypeError swallowed -> LU (container SciPy behaviour, F2)
    run("gmres_legacy", 32, 2, 91, "iterative_gmres", 10.0, False)
    return arrays, {"versions": versions(), "cases": meta}


def nan_ladder(mod):
    """G2: NaN-poisoned 8x8 eig problem -- every step is a total failure."""
    n = 8
    A = scenarios.ginibre(n, 55, 1.0)
    A[3, 4] = np.nan
    seed_all(mod, 17)
    S = mod.SolutionCandidate.State
    c = mod.SolutionCandidate(A, mod.ProblemType.EIGENVALUE, n)
    strat = {"overall_psi_aggression_factor": 1.0, "max_psi_retries": 25, "current_convergence_threshold": 1e-8,
             "convergence_tolerance": 1e-8}
    know = {"local_solver_preference": "direct_solve", "is_sparse_problem": False, "is_hermitian": False}
    steps = []
    for _ in range(10):
        if c.state in (S.CONVERGED, S.RETIRED):
            # the evolve loop would skip it; keep stepping anyway to pin the fall-through behaviour
            pass
        c.update_solution_step(A, None, strat, know)
        a = complex(c.alpha_local_step)
        steps.append({"state": c.state.value, "stuck": c.stuck_counter, "retries": c.local_psi_retries_needed,
                      "resets": c.num_resets, "w": float(c.w_k).hex(), "alpha": [a.real.hex(), a.imag.hex()],
                      "resid_nan": bool(np.isnan(c.residual_k)), "hist_len": len(c.residual_history),
                      "rng": snapshot.rng_digest(), "mt_pos": int(np.random.get_state()[2])})
    return {"versions": versions(), "n": n, "seed": 17, "matrix_seed": 55, "nan_at": [3, 4], "steps": steps}


def main():
    mod = load_reference()
    out_dir = HERE
    for name, spec in scenarios.TRAJECTORIES.items():
        rec = run_trajectory(mod, name, spec)
        with open(os.path.join(out_dir, f"traj_{name}.json"), "w") as f:
            json.dump(rec, f, indent=0, separators=(",", ":"))
        print(name, "steps", rec["total_steps"], "final pop", rec["iters"][-1]["n_after"])
    arrays, meta = solve_fixtures(mod)
    np.savez_compressed(os.path.join(out_dir, "solve_cases.npz"), **arrays)
    with open(os.path.join(out_dir, "solve_cases.json"), "w") as f:
        json.dump(meta, f, indent=1)
    with open(os.path.join(out_dir, "nan_ladder.json"), "w") as f:
        json.dump(nan_ladder(mod), f, indent=1)
    print("done")


if __name__ == "__main__":
    main()
