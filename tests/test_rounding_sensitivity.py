"""CPU evidence for the tolerances and horizons that the long GPU trajectory tests grant the device.

Three long scenarios failed step-by-step parity on the GPU in round 2 and were loosened:
  * lap8_p96 (tests/test_gpu_bench_path.py): |dlambda| 2e-9 at iteration 32, 2.6e-8 at 37 -> tolerance x 30, 34 iterations;
  * lap8 through evolve() (tests/test_gpu_evolve.py): 60 iterations -> 25;
  * eig288_p64: "population after manage differs" at iteration 13 -> converged survivors compared as a multiset.
The claim behind the loosening was that these are rounding effects, not bugs.  Here the ORACLE is run against ITSELF with
the matrix of every solve perturbed by one unit in the last place (tests/rounding.py): if a backward-error-sized change
on the reference side alone produces the same divergence at the same iterations, the claim holds, and the numbers the GPU
tests use are derived from these measurements instead of being picked to make a run pass."""
import numpy as np
import pytest

import rounding
import scenarios
from oracle import maus_oracle as orc

LAP8_P96 = dict(kind="eig", build=("laplace", 8, 8, False), P=96, iters=60, seed=7, tol=1e-7)


@pytest.fixture
def lap8_p96():
    scenarios.TRAJECTORIES["lap8_p96"] = LAP8_P96
    yield "lap8_p96"
    scenarios.TRAJECTORIES.pop("lap8_p96", None)


def test_one_ulp_in_H_reproduces_the_lap8_p96_drift_seen_on_the_gpu(lap8_p96):
    iters = 60
    env, first_int, first_order = rounding.envelope(lap8_p96, iters, seeds=(1, 2, 3))
    # (i) the drift grows geometrically from rounding level: 1e-16 -> 1e-9 in ~32 iterations (the GPU run: 1.7x per iteration)
    lo, hi = 5, 30
    rate = (env[hi] / env[lo]) ** (1.0 / (hi - lo))
    assert 1.3 < rate < 2.2, rate
    # (ii) it crosses the base tolerance of the step-parity tests (1e-9) where the GPU run did: iteration 32 (2e-9 there)
    first = int(np.argmax(env > 1e-9))
    assert 29 <= first <= 35, (first, env[28:38])
    assert 2e-10 < env[32] < 2e-8 and 1e-8 < env[37] < 1e-6, (env[32], env[37])
    # (iii) bookkeeping (states, counters, both RNG streams) is untouched for a long time after that, the survivor order
    #       goes first, integer bookkeeping later: step-by-step parity is meaningful up to ~45 iterations, not beyond
    assert first_order >= 40 and first_int >= first_order, (first_order, first_int)
    assert first_order < iters, "a 60-iteration run does diverge: the old 60-iteration GPU comparison could not pass"
    # (iv) what tests/test_gpu_bench_path.py grants the device over its 34 iterations follows from this envelope
    scale = rounding.granted_scale(env, 1e-9)
    assert scale[:25].max() == 1.0, "no loosening where a backward-error-sized perturbation stays below the base tolerance"
    assert 4.0 < scale[33] < 200.0, scale[33]


def test_lap8_evolve_horizon_follows_from_the_perturbed_oracle():
    """tests/test_gpu_evolve.py compares the end state of evolve() on `lap8` (30 candidates).  With one ulp in H the oracle
    itself keeps its bookkeeping and survivor order for 45+ iterations and loses them before 60; the evolve test runs
    EVOLVE_LAP8_ITERS iterations, inside that window with DEVICE_ULPS of margin (1.65^4 ~ 8: four iterations earlier)."""
    from rounding import EVOLVE_LAP8_ITERS
    env, first_int, first_order = rounding.envelope("lap8", 60, seeds=(1, 2, 3))
    assert 44 <= first_order < 60, first_order
    assert first_int >= first_order
    assert EVOLVE_LAP8_ITERS <= first_order - 5
    # the eigenvalues the evolve test compares at 1e-9 are those of CONVERGED candidates (pinned by their residual, they do
    # not drift); the active ones enter through AvgRes, compared at 1e-2: their drift is orders of magnitude below that
    assert rounding.DEVICE_ULPS * env[EVOLVE_LAP8_ITERS - 1] < 1e-4


def test_duplicate_eigenpairs_are_retired_by_rounding_noise():
    """AMS:506 sorts by (-w_k, residual_k) and AMS:509-520 retire the LATER ones of several converged duplicates.
    Converged candidates all have w_k = 1 and residuals that are rounding noise of ||A v - lambda v||, so which duplicate
    survives is decided by the last bits of the solves -- `eig288_p64 iter 13: population after manage differs` on the GPU,
    and at iteration 13-14 of the same scenario on the CPU when the oracle is perturbed by one ulp (tools/
    rounding_sensitivity.py, profiles/r03_rounding_sensitivity.txt).  The mechanism at its core, and cheap: ten candidates
    converging to the same eigenpair of that 288 x 288 matrix.  With one ulp in H their residuals stay in the noise band
    and their sort ORDER -- hence the survivor -- changes from one perturbation seed to the next."""
    n, ncand = 288, 10
    A = scenarios.ginibre(n, 300, None)
    anorm = np.linalg.norm(A, 1)
    lam, V = np.linalg.eig(A)
    k = int(np.argmax(lam.real))
    strat = {"overall_psi_aggression_factor": 1.0, "max_psi_retries": 25, "current_convergence_threshold": 1e-8,
             "convergence_tolerance": 1e-8}
    know = {"local_solver_preference": orc.DIRECT, "is_sparse_problem": False, "is_hermitian": False}

    def family(seed):
        orc.seed_all(99)
        rng = np.random.default_rng(5)
        cands = []
        for j in range(ncand):
            c = orc.new_candidate(A, orc.EIGENVALUE, n)
            v = V[:, k] + 1e-4 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
            c.v = v / np.linalg.norm(v)
            c.lam = lam[k]
            c.alpha = 1.0
            cands.append(c)
        real = orc.sla
        if seed is not None:
            orc.sla = rounding._PerturbedSolve(real, seed, 1)
        try:
            for _ in range(6):
                for c in cands:
                    if c.state != orc.CONVERGED:
                        orc.candidate_step(c, A, None, strat, know)
        finally:
            orc.sla = real
        assert all(c.state == orc.CONVERGED and c.w == 1.0 for c in cands)
        for c in cands[1:]:                                                        # duplicates by the rule of AMS:435-436
            assert abs(c.lam - cands[0].lam) < 1e-5 and abs(np.vdot(c.v, cands[0].v)) > 0.999
        resid = np.array([c.resid for c in cands])
        noise = resid < 1e-11 * anorm               # (one of the ten crosses the 1e-8 threshold a step early, at 7e-9)
        assert noise.sum() >= 8, "converged residuals are rounding noise"
        # the survivor of manage_candidates: first in the stable sort by (-w, residual)
        return tuple(int(j) for j in np.argsort(resid, kind="stable") if noise[j])

    orders = {family(seed) for seed in [None] + list(range(1, 9))}
    assert len(orders) >= 2, "the sort order of converged duplicates did not change under 1-ulp perturbations"
    assert len({o[0] for o in orders}) >= 1
