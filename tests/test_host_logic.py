"""The product's HOST logic (solver.py + engine.py) against the fixtures captured from the
reference, on CPU: the device phases are replaced by the NumPy test double tests/fake_ctx.py,
so everything that must be bit-exact -- both RNG streams, the retry/fallback ladder, alpha /
state machine, diagnostics, strategy, retire/spawn -- is checked digest by digest."""
import json
import os
import random

import numpy as np
import pytest

import scenarios
import snapshot
from fake_ctx import FakeContext

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def make_solver(name, **kw):
    from adaptive_matrix_solver_amd.engine import DeviceEngine
    from adaptive_matrix_solver_amd.solver import MAUS_Solver, ProblemType, SolutionCandidate
    spec = scenarios.TRAJECTORIES[name]
    A, b = scenarios.build(spec)
    np.random.seed(spec["seed"]); random.seed(spec["seed"]); SolutionCandidate._candidate_id_counter = 0
    PT = {"eig": ProblemType.EIGENVALUE, "lin": ProblemType.SOLVE_LINEAR_SYSTEM, "svd": ProblemType.SVD}[spec["kind"]]
    eng = DeviceEngine(ctx=FakeContext(), pert_mode=kw.pop("pert_mode", "uniform"),
                       gmres_compat=kw.pop("gmres_compat", "scipy-legacy"), comm=kw.pop("comm", None))
    return MAUS_Solver(A, PT, b_vector=b, initial_num_candidates=spec["P"], global_convergence_tol=spec["tol"],
                       quiet=True, engine=eng, record_history=True), spec


def rows_of(cands, kind):
    rows = []
    for c in cands:
        if kind == "eig":
            lam, vecs = c.lambda_k, [c.v_k]
        elif kind == "lin":
            lam, vecs = 0j, [c.x_k]
        else:
            lam, vecs = c.sigma_k, [c.u_k, c.right_v_k]
        rows.append({"id": c.id, "state": c.state.value, "stuck": c.stuck_counter, "retries": c.local_psi_retries_needed,
                     "resets": c.num_resets, "w": c.w_k, "resid": c.residual_k, "alpha": c.alpha_local_step,
                     "lam": lam, "vecs": vecs})
    return rows


def versions_match(rec):
    import scipy
    return rec["versions"]["numpy"] == np.__version__ and rec["versions"]["scipy"] == scipy.__version__


# (the Hermitian scenarios are bit-exact too: eigh is computed once per matrix instead of once per
# candidate, but it is the same LAPACK call on the same input, so (lambda, V) are identical)
@pytest.mark.parametrize("name", ["eig16", "eig64", "eig48u", "lap8", "lin24", "lin32f", "svd5x4", "svd64", "svdtiny",
                                  "herm16", "herm64", "lap8h"])
def test_host_logic_bit_exact_vs_reference_fixtures(name):
    with open(os.path.join(GOLD, f"traj_{name}.json")) as f:
        gold = json.load(f)
    if not versions_match(gold):
        pytest.skip("fixture captured under different numpy/scipy versions")
    solver, spec = make_solver(name)
    assert snapshot.digest_rows(rows_of(solver.candidates, spec["kind"])) == gold["init"]["digest"]
    assert snapshot.rng_digest() == gold["init"]["rng"]
    from adaptive_matrix_solver_amd.solver import SolutionCandidate
    for it, g in enumerate(gold["iters"]):
        solver._update_global_diagnostics(it + 1)
        solver._adjust_global_strategy(it + 1)
        steps = solver.step_population()
        stepped = rows_of(solver.candidates, spec["kind"])
        solver._manage_candidates(it + 1)
        assert steps == g["steps"], f"iter {it}"
        d = snapshot.digest_rows(stepped)
        assert d["ints"] == g["digest_stepped"]["ints"], f"iter {it} bookkeeping"
        assert snapshot.rng_digest() == g["rng"], f"iter {it} rng"
        assert d["floats"] == g["digest_stepped"]["floats"], f"iter {it} scalars"
        assert d["vecs"] == g["digest_stepped"]["vecs"], f"iter {it} vectors"
        assert snapshot.digest_rows(rows_of(solver.candidates, spec["kind"])) == g["digest"], f"iter {it} managed"
        got_glob = snapshot.globals_record(solver.landscape_energy, solver.avg_residual, solver.avg_stuckness,
                                           solver.num_distinct_converged_solutions,
                                           solver.problem_knowledge["numerical_stability_state"],
                                           solver.problem_knowledge["local_solver_preference"], solver.strat_params)
        assert got_glob == g["globals"], f"iter {it} globals"
        assert SolutionCandidate._candidate_id_counter == g["next_id"]


def test_fast_mode_keeps_stream_and_bookkeeping():
    """pert_mode='none': the perturbation is dropped and the NumPy stream is advanced by the
    MT19937 jump -- stream position and bookkeeping must still equal the reference's."""
    with open(os.path.join(GOLD, "traj_eig64.json")) as f:
        gold = json.load(f)
    if not versions_match(gold):
        pytest.skip("fixture captured under different numpy/scipy versions")
    solver, spec = make_solver("eig64", pert_mode="none")
    for it, g in enumerate(gold["iters"]):
        solver._update_global_diagnostics(it + 1)
        solver._adjust_global_strategy(it + 1)
        solver.step_population()
        stepped = rows_of(solver.candidates, spec["kind"])
        solver._manage_candidates(it + 1)
        assert snapshot.digest_rows(stepped)["ints"] == g["digest_stepped"]["ints"], f"iter {it}"
        assert snapshot.rng_digest() == g["rng"], f"iter {it} rng"


def test_failed_gmres_attempts_are_retried_as_one_direct_batch(monkeypatch):
    """AMS:99-103 in batch (engine._solve): GMRES preferred, stream-independent host side ('none' here, 'mt19937' on the
    device), attempt 0 fails for some candidates -> ONE batched direct solve for exactly those, at the same attempt index; the
    NumPy stream moves by one rand(N,N) pair per attempt (the failed candidates consumed two)."""
    from adaptive_matrix_solver_amd.engine import GMRES
    solver, spec = make_solver("lin24", pert_mode="none", gmres_compat="rtol")
    solver.problem_knowledge["local_solver_preference"] = GMRES
    ctx = solver.engine.ctx
    real_gmres, real_lu = ctx.gmres, ctx.shifted_lu_solve
    lu_batches, fails = [], []

    def flaky_gmres(slots, shift, psi, rhs_mode, use_j, **kw):
        info, inner, status = real_gmres(slots, shift, psi, rhs_mode, use_j, **kw)
        info = np.array(info)
        info[::3] = 1                                           # every third attempt "does not converge"
        fails.append(int(np.count_nonzero((info != 0) | (np.asarray(status) != 0))))
        return info, inner, status

    def counting_lu(slots, *a, **kw):
        lu_batches.append(len(slots))
        return real_lu(slots, *a, **kw)
    monkeypatch.setattr(ctx, "gmres", flaky_gmres)
    monkeypatch.setattr(ctx, "shifted_lu_solve", counting_lu)
    n = solver.N_diag
    active = list(solver.candidates)
    before = np.random.get_state()
    solver.step_population()
    assert len(fails) == 1 and fails[0] >= len(range(0, len(active), 3))
    assert lu_batches == fails                                  # one direct batch, the failed candidates only
    assert all(c.local_psi_retries_needed == 0 for c in active)
    assert all(np.isfinite(c.residual_k) for c in active)
    after = snapshot.rng_digest()
    np.random.set_state(before)
    for _ in range(len(active) + fails[0]):                     # (candidates + failures) x two rand(n, n)
        np.random.rand(n, n); np.random.rand(n, n)
    assert snapshot.rng_digest() == after


def test_nan_ladder_matches_reference():
    from adaptive_matrix_solver_amd.engine import DeviceEngine
    from adaptive_matrix_solver_amd.solver import ProblemType, SolutionCandidate
    with open(os.path.join(GOLD, "nan_ladder.json")) as f:
        gold = json.load(f)
    n = gold["n"]
    A = scenarios.ginibre(n, gold["matrix_seed"], 1.0)
    A[tuple(gold["nan_at"])] = np.nan
    np.random.seed(gold["seed"]); random.seed(gold["seed"]); SolutionCandidate._candidate_id_counter = 0
    eng = DeviceEngine(ctx=FakeContext(), pert_mode="uniform", gmres_compat="scipy-legacy")
    c = SolutionCandidate(A, ProblemType.EIGENVALUE, n, engine=eng)
    strat = {"overall_psi_aggression_factor": 1.0, "max_psi_retries": 25, "current_convergence_threshold": 1e-8,
             "convergence_tolerance": 1e-8}
    know = {"local_solver_preference": "direct_solve", "is_sparse_problem": False, "is_hermitian": False}
    for i, g in enumerate(gold["steps"]):
        with np.errstate(all="ignore"):
            c.update_solution_step(A, None, strat, know)
        a = complex(c.alpha_local_step)
        got = {"state": c.state.value, "stuck": c.stuck_counter, "retries": c.local_psi_retries_needed,
               "resets": c.num_resets, "w": float(c.w_k).hex(), "alpha": [a.real.hex(), a.imag.hex()],
               "resid_nan": bool(np.isnan(c.residual_k)), "hist_len": len(c.residual_history),
               "rng": snapshot.rng_digest(), "mt_pos": int(np.random.get_state()[2])}
        assert got == g, f"step {i}"


def test_sparse_inputs_rejected_loudly():
    import scipy.sparse as sp
    from adaptive_matrix_solver_amd.solver import MAUS_Solver, ProblemType
    with pytest.raises(NotImplementedError):
        MAUS_Solver(sp.identity(8, format="csc"), ProblemType.EIGENVALUE)


def test_mt19937_jump_equals_drawing():
    from adaptive_matrix_solver_amd import _cabi
    for seed, nwords, pre in [(1, 100, 0), (2, 4 * 64 * 64, 17), (3, 4 * 300 * 300, 623), (4, 4 * 1024 * 1024, 5)]:
        np.random.seed(seed)
        if pre:
            np.random.rand(pre)
        st = np.random.get_state()
        key, pos = _cabi.mt19937_jump(st[1], st[2], nwords)
        np.random.rand(nwords // 2)
        st2 = np.random.get_state()
        assert np.array_equal(key, st2[1]) and pos == st2[2]


def test_evolve_reports_like_the_reference(capsys):
    """evolve(): loop, final report with recomputed residuals, and the comparison with the SciPy answer (AMS:551-608;
    F1 fixed).  Hermitian 16x16: every candidate converges in the first step."""
    import random
    from adaptive_matrix_solver_amd.engine import DeviceEngine
    from adaptive_matrix_solver_amd.solver import MAUS_Solver, ProblemType, SolutionCandidate
    A = scenarios.hermitian(16, 16)
    np.random.seed(5)
    random.seed(5)
    SolutionCandidate._candidate_id_counter = 0
    eng = DeviceEngine(ctx=FakeContext(), pert_mode="uniform")
    solver = MAUS_Solver(A, ProblemType.EIGENVALUE, initial_num_candidates=12, global_convergence_tol=1e-8, quiet=True, engine=eng)
    solver.evolve(max_iterations=4)
    out = capsys.readouterr().out
    assert "--- Starting MAUS Evolution for 4 iterations (EIGENVALUE) ---" in out
    assert "Final Report:" in out and "  Eig 1: λ=" in out and ", Res=" in out
    assert "--- Comparison to NumPy ---" in out
    line = [l for l in out.splitlines() if l.startswith("Mean abs error (eigs):")]
    assert line and float(line[0].split(":")[1]) < 10.0        # sorted-prefix comparison, as the reference prints it
    assert solver.true_solution is not None and len(solver.true_solution) == 16
    # every reported eigenpair is one: residuals at rounding level
    res = [float(l.split("Res=")[1]) for l in out.splitlines() if l.startswith("  Eig ")]
    assert res and max(res) < 1e-10


FLOAT_ALPHA = {"lap8", "eig48u", "svd5x4"}


@pytest.mark.parametrize("name,iters", [("eig16", 10), ("lap8", 40), ("eig48u", 26), ("lin24", 8), ("svd5x4", 10)])
def test_alpha_local_step_keeps_the_reference_types(name, iters):
    """AMS:124, 308-314, 331: alpha is np.complex128 until a clamp returns the Python-float bound or
    convergence assigns 0.0.  The oracle keeps the reference's expressions verbatim, so its types are the reference's; compared
    as type(), not as complex() -- which is blind to the difference."""
    from oracle import maus_oracle as orc
    spec = scenarios.TRAJECTORIES[name]
    A, b = scenarios.build(spec)
    orc.seed_all(spec["seed"])
    kind = {"eig": orc.EIGENVALUE, "lin": orc.SOLVE_LINEAR_SYSTEM, "svd": orc.SVD}[spec["kind"]]
    pop = orc.new_population(A, kind, b=b, n_cands=spec["P"], tol=spec["tol"])
    ref = []
    for _ in range(iters):
        orc.update_diagnostics(pop)
        orc.adjust_strategy(pop)
        for c in pop.cands:
            if c.state not in (orc.CONVERGED, orc.RETIRED):
                orc.candidate_step(c, pop.M, pop.b, pop.strat, pop.know, gmres_mode="scipy-legacy")
        ref.append([(c.cid, type(c.alpha), complex(c.alpha)) for c in pop.cands])
        orc.manage_candidates(pop)
    solver, _ = make_solver(name)
    seen = set()
    for it in range(iters):
        solver._update_global_diagnostics(it + 1)
        solver._adjust_global_strategy(it + 1)
        solver.step_population()
        got = [(c.id, type(c.alpha_local_step), complex(c.alpha_local_step)) for c in solver.candidates]
        assert got == ref[it], f"iteration {it + 1}"
        seen |= {t for _, t, _ in got}
        solver._manage_candidates(it + 1)
    assert np.complex128 in seen
    if name in FLOAT_ALPHA:
        assert float in seen            # a clamp or a convergence happened: the comparison above saw both kinds


def test_general_eigenvalue_prologue_is_lazy(monkeypatch, capsys):
    """AMS:559: the O(n^3) host eigvals of evolve()'s prologue runs when `true_solution` is first read (closing comparison, or
    the caller), not in front of the first iteration; small matrices keep the reference's order."""
    from adaptive_matrix_solver_amd import solver as sv
    calls = []
    real = sv.MAUS_Solver._reference_solution
    monkeypatch.setattr(sv.MAUS_Solver, "_reference_solution", lambda self: (calls.append(len(self.candidates[0].residual_history)), real(self))[1])
    solver, _ = make_solver("lap8")
    solver.evolve(3)                                               # n = 8 <= REFERENCE_LAZY_MIN: eager, before any step
    assert calls == [1]
    monkeypatch.setattr(sv, "REFERENCE_LAZY_MIN", 4)
    calls.clear()
    solver, _ = make_solver("lap8")
    solver.evolve(3)
    if solver.num_distinct_converged_solutions == 0:
        assert calls == []                                         # nothing to compare with: never computed ...
    ref = solver.true_solution                                     # ... until somebody reads it
    assert len(calls) == 1 and calls[0] > 1
    assert np.allclose(np.sort_complex(ref), np.sort_complex(np.linalg.eigvals(solver.M)))
    assert solver.true_solution is ref and len(calls) == 1
    capsys.readouterr()
