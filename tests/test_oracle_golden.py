"""Pin the CPU oracle (oracle/maus_oracle.py) against fixtures captured from the
reference itself (tests/golden/make_goldens.py).  Bit-exact: the oracle makes
the same NumPy/SciPy/LAPACK calls in the same order and consumes both RNG
streams identically, so every digest must match.  CPU only."""
import json
import os

import numpy as np
import pytest

import scenarios
import snapshot
from oracle import maus_oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KIND = {"eig": orc.EIGENVALUE, "lin": orc.SOLVE_LINEAR_SYSTEM, "svd": orc.SVD}


def _versions_match(rec):
    import scipy
    return rec["versions"]["numpy"] == np.__version__ and rec["versions"]["scipy"] == scipy.__version__


def rows_of(pop_cands, kind):
    rows = []
    for c in pop_cands:
        if kind == "eig":
            lam, vecs = c.lam, [c.v]
        elif kind == "lin":
            lam, vecs = 0j, [c.x]
        else:
            lam, vecs = c.sigma, [c.u, c.v]
        rows.append({"id": c.cid, "state": c.state, "stuck": c.stuck, "retries": c.retries,
                     "resets": c.resets, "w": c.w, "resid": c.resid, "alpha": c.alpha, "lam": lam,
                     "vecs": vecs})
    return rows


def globals_of(pop):
    return snapshot.globals_record(pop.energy, pop.avg_resid, pop.avg_stuck, pop.n_distinct,
                                   pop.know["numerical_stability_state"],
                                   pop.know["local_solver_preference"], pop.strat)


@pytest.mark.parametrize("name", sorted(scenarios.TRAJECTORIES))
def test_trajectory_bit_exact(name):
    with open(os.path.join(GOLD, f"traj_{name}.json")) as f:
        gold = json.load(f)
    if not _versions_match(gold):
        pytest.skip("fixture captured under different numpy/scipy versions")
    spec = scenarios.TRAJECTORIES[name]
    A, b = scenarios.build(spec)
    orc.seed_all(spec["seed"])
    pop = orc.new_population(A, KIND[spec["kind"]], b=b, n_cands=spec["P"], tol=spec["tol"])
    assert float(pop.diag["condition_number"]).hex() == gold["cond"]
    assert pop.know["is_hermitian"] == gold["hermitian"]
    assert snapshot.digest_rows(rows_of(pop.cands, spec["kind"])) == gold["init"]["digest"]
    assert snapshot.rng_digest() == gold["init"]["rng"]
    total = 0
    for it, g in enumerate(gold["iters"]):
        orc.update_diagnostics(pop)
        orc.adjust_strategy(pop)
        steps = 0
        for c in pop.cands:
            if c.state not in (orc.CONVERGED, orc.RETIRED):
                orc.candidate_step(c, pop.M, pop.b, pop.strat, pop.know)
                steps += 1
        stepped = rows_of(pop.cands, spec["kind"])
        orc.manage_candidates(pop)
        total += steps
        assert steps == g["steps"], f"iter {it}"
        if "rows" in g:
            assert snapshot.full_rows(stepped) == g["rows"], f"iter {it} rows"
        assert snapshot.digest_rows(stepped) == g["digest_stepped"], f"iter {it} stepped"
        assert snapshot.digest_rows(rows_of(pop.cands, spec["kind"])) == g["digest"], f"iter {it} managed"
        assert snapshot.rng_digest() == g["rng"], f"iter {it} rng"
        assert globals_of(pop) == g["globals"], f"iter {it} globals"
        assert orc.IdCounter.value == g["next_id"]
    assert total == gold["total_steps"]


def test_solve_cases():
    with open(os.path.join(GOLD, "solve_cases.json")) as f:
        meta = json.load(f)
    arrays = np.load(os.path.join(GOLD, "solve_cases.npz"))
    exact = _versions_match(meta)
    for case in meta["cases"]:
        tgt, rhs = scenarios.solve_case_inputs(case["key"])
        orc.seed_all(case["seed"])
        mode = "rtol" if case["gmres_shim"] else "scipy-legacy"
        trace = []
        x, att = orc.inverse_iterate_solve(tgt, rhs, case["stuck"], n=case["n"],
                                           base_psi=orc.PSI_EPSILON_BASE * case["aggr"],
                                           max_attempts=case["max_attempts"], preferred=case["pref"],
                                           gmres_mode=mode, trace=trace)
        assert att == case["attempts"], case["key"]
        assert snapshot.rng_digest() == case["rng"], case["key"]
        gx = arrays[case["key"] + "_x"]
        if exact:
            assert np.array_equal(x, gx), case["key"]
        else:
            assert np.allclose(x, gx, rtol=1e-10, atol=0)
        # MT19937 consumption: 4*n*n words per dense attempt
        n_attempts = len(trace)
        words = 4 * case["n"] ** 2 * n_attempts
        assert (case["mt_pos"][0] + words) % 624 == case["mt_pos"][1] % 624, case["key"]
        if case["key"] == "gmresfb_n32_s0":
            assert [t["method"] for t in trace] == [orc.GMRES, orc.DIRECT]
        if case["key"] == "gmres_legacy":
            assert [t["method"] for t in trace] == [orc.GMRES, orc.DIRECT]


def test_gmres_restated_matches_scipy():
    """The oracle's own GMRES restatement (what the HIP kernel implements) against
    SciPy's gmres on the fixture systems: same iterate to rounding, same info."""
    for key, stuck in (("gmres_n32_s0", 0), ("gmres_n32_s2", 2), ("gmres_n128_s0", 0),
                       ("gmres_n128_s2", 2), ("gmresfb_n32_s0", 0)):
        A, b = scenarios.solve_case_inputs(key)
        inv_d = orc.jacobi_inverse_diagonal(A, stuck)
        xs, info_s = orc.gmres_scipy(A, b, b, inv_d)
        xr, info_r, inner, cycles = orc.gmres_restated(A, b, b, inv_d)
        assert (info_s == 0) == (info_r == 0), key
        assert info_s == info_r, key
        if info_s == 0:
            assert np.linalg.norm(xs - xr) <= 1e-9 * np.linalg.norm(xs), key
            assert inner >= 1 and cycles >= 1
        else:
            assert inner == 50 * 20 and cycles == 50


def test_zlartg_against_lapack():
    from scipy.linalg import get_lapack_funcs
    lartg = get_lapack_funcs("lartg", dtype=np.complex128)
    rng = np.random.default_rng(0)
    for _ in range(200):
        f = complex(rng.standard_normal(), rng.standard_normal()) * 10.0 ** rng.integers(-5, 5)
        g = complex(abs(rng.standard_normal())) * 10.0 ** rng.integers(-5, 5)
        c0, s0, r0 = lartg(f, g)
        c1, s1, r1 = orc.zlartg(f, g)
        assert abs(c0 - c1) <= 4e-16 * max(1.0, abs(c0))
        assert abs(s0 - s1) <= 4e-16 * max(1.0, abs(s0))
        assert abs(r0 - r1) <= 4e-16 * abs(r0)
    assert orc.zlartg(1 + 2j, 0) == (1.0, 0j, 1 + 2j)
    c, s, r = orc.zlartg(0, 3.0)
    assert c == 0.0 and s == 1.0 and r == 3.0


def test_nan_ladder():
    with open(os.path.join(GOLD, "nan_ladder.json")) as f:
        gold = json.load(f)
    n = gold["n"]
    A = scenarios.ginibre(n, gold["matrix_seed"], 1.0)
    A[tuple(gold["nan_at"])] = np.nan
    orc.seed_all(gold["seed"])
    c = orc.new_candidate(A, orc.EIGENVALUE, n)
    strat = {"overall_psi_aggression_factor": 1.0, "max_psi_retries": 25,
             "current_convergence_threshold": 1e-8, "convergence_tolerance": 1e-8}
    know = {"local_solver_preference": orc.DIRECT, "is_sparse_problem": False, "is_hermitian": False}
    prev_pos = None
    for i, g in enumerate(gold["steps"]):
        with np.errstate(all="ignore"):
            orc.candidate_step(c, A, None, strat, know)
        a = complex(c.alpha)
        got = {"state": c.state, "stuck": c.stuck, "retries": c.retries, "resets": c.resets,
               "w": float(c.w).hex(), "alpha": [a.real.hex(), a.imag.hex()],
               "resid_nan": bool(np.isnan(c.resid)), "hist_len": len(c.resid_hist),
               "rng": snapshot.rng_digest(), "mt_pos": int(np.random.get_state()[2])}
        assert got == g, f"step {i}"
    # SURVEY appendix B known answer: STUCK for steps 1-7, RETIRED from step 8
    assert [s["state"] for s in gold["steps"][:8]] == [orc.STUCK] * 7 + [orc.RETIRED]


def test_psi_schedule_types():
    p = orc.psi_magnitude(orc.PSI_EPSILON_BASE * 1.0, 3, 2)
    assert isinstance(p, np.complex128) and p.imag == 0.0
    assert p.real == (1e-20 * (10 ** 1.5)) * (10 ** (2 / 3.0))
