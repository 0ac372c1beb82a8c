"""TEST INFRASTRUCTURE: how far does the ORACLE move when the matrix it factorises is perturbed at rounding level?

The device LU and LAPACK's zgetrf are both backward stable: each returns the exact solution of a system whose matrix
differs from H by a few ulps per entry -- not the same few ulps.  To tell "the device differs from the oracle because it
rounds differently" from "the device has a bug", the oracle is run against itself with every entry of every H_solve moved
by up to `ulps` units in the last place before scipy.linalg.solve (AMS:59) sees it.  Everything else -- both RNG streams,
the retry ladder, bookkeeping -- is untouched, so the two runs consume the same random numbers and differ only through
that backward-error-sized perturbation.  Used by tests/test_rounding_sensitivity.py (CPU) to pin what such a
perturbation does to the long trajectories, and by tests/test_gpu_bench_path.py / test_gpu_evolve.py to derive the
tolerance and the horizon they grant the device."""
import numpy as np

from oracle import maus_oracle as orc


class _PerturbedSolve:
    """Stands in for the oracle module's `sla`: solve() perturbs H by up to `ulps` ulps per real / imaginary part (private
    generator: the global legacy streams are bookkeeping), everything else is forwarded."""

    def __init__(self, real, seed, ulps):
        self._real, self._rng, self._ulps = real, np.random.default_rng(seed), int(ulps)

    def solve(self, H, b, **kw):
        r = self._rng.integers(-self._ulps, self._ulps + 1, size=(2,) + H.shape)
        H2 = (H.real + r[0] * np.spacing(H.real)) + 1j * (H.imag + r[1] * np.spacing(H.imag))
        return self._real.solve(H2, b, **kw)

    def __getattr__(self, name):
        return getattr(self._real, name)


def oracle_run(name, iters, perturb_seed=None, ulps=1, **kw):
    """test_gpu_step_parity.oracle_run, optionally with the rounding-level perturbation."""
    import test_gpu_step_parity as sp
    real = orc.sla
    if perturb_seed is not None:
        orc.sla = _PerturbedSolve(real, perturb_seed, ulps)
    try:
        return sp.oracle_run(name, iters, **kw)
    finally:
        orc.sla = real


def drift(ref, got):
    """Per iteration: (max |dlambda| / max(1, |lambda|) over the candidates stepped in both runs, integer bookkeeping equal,
    survivor list equal, RNG digests equal)."""
    out = []
    for r, g in zip(ref, got):
        gb = {x["id"]: x for x in g["rows"]}
        d, same = 0.0, len(r["rows"]) == len(g["rows"])
        for x in r["rows"]:
            y = gb.get(x["id"])
            if y is None:
                same = False
                continue
            if (x["state"], x["stuck"], x["retries"], x["resets"]) != (y["state"], y["stuck"], y["retries"], y["resets"]):
                same = False
            if x["state"] not in (orc.CONVERGED, orc.RETIRED) and y["state"] not in (orc.CONVERGED, orc.RETIRED):
                d = max(d, abs(x["lam"] - y["lam"]) / max(1.0, abs(x["lam"])))
        out.append((d, same, r["after"] == g["after"], r["rng"] == g["rng"]))
    return out


def envelope(name, iters, seeds=(1, 2, 3), ulps=1, ref=None):
    """max over `seeds` of the per-iteration eigenvalue drift of the perturbed oracle against the unperturbed one, and the
    first iteration at which any seed's bookkeeping / survivor order left the reference's."""
    if ref is None:
        ref, _ = oracle_run(name, iters)
    env = np.zeros(iters)
    first_int, first_order = iters, iters
    for s in seeds:
        got, _ = oracle_run(name, iters, perturb_seed=s, ulps=ulps)
        for it, (d, same, order, _) in enumerate(drift(ref, got)):
            env[it] = max(env[it], d)
            if not same:
                first_int = min(first_int, it)
            if not order:
                first_order = min(first_order, it)
    return env, first_int, first_order


# Backward error of an LU with partial pivoting in units of ulp(H): what the device is granted relative to the 1-ulp
# perturbation above.  The textbook bound is ~ n * growth; measured backward errors of zgetrf and of the device LU are a
# few ulps.  8 covers both sides of the comparison (LAPACK's own error and the device's).
DEVICE_ULPS = 8.0


def granted_scale(env, base_tol):
    """Per-iteration multiplier of a base eigenvalue tolerance: 1 while a DEVICE_ULPS-sized backward perturbation stays
    below the tolerance, beyond that what such a perturbation measurably produces (with a factor 2 of head room for the
    spread between perturbation seeds)."""
    return np.maximum(1.0, 2.0 * DEVICE_ULPS * np.asarray(env) / base_tol)


# Horizon of the `lap8` evolve() comparison on the GPU (tests/test_gpu_evolve.py), justified on the CPU by
# tests/test_rounding_sensitivity.py::test_lap8_evolve_horizon_follows_from_the_perturbed_oracle: the oracle perturbed by one
# ulp per entry of H keeps bookkeeping and survivor order for >= 45 iterations; 8 ulps cost ~4 iterations of that.
EVOLVE_LAP8_ITERS = 40
