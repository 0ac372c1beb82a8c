"""Canonical digests of a population state, used on both sides of the golden
comparison (reference objects in make_goldens.py, oracle/product records in the
tests).  A "row" is a plain dict per candidate, in list order:

  id, state, stuck, retries, resets : int
  w, resid                          : float
  alpha, lam                        : complex (lam: lambda_k or sigma_k or 0)
  vecs                              : list of complex128 ndarrays (v | x | u,v)
"""
import hashlib
import random

import numpy as np


def _f(x):
    return float(x).hex()


def _c(z):
    z = complex(z)
    return [z.real.hex(), z.imag.hex()]


def digest_rows(rows):
    hi = hashlib.sha256()
    hf = hashlib.sha256()
    hv = hashlib.sha256()
    for r in rows:
        hi.update(np.array([r["id"], r["state"], r["stuck"], r["retries"], r["resets"]], dtype=np.int64).tobytes())
        a = complex(r["alpha"])
        l = complex(r["lam"]) if r["lam"] is not None else 0j
        hf.update(np.array([r["w"], r["resid"], a.real, a.imag, l.real, l.imag], dtype=np.float64).tobytes())
        for v in r["vecs"]:
            hv.update(np.ascontiguousarray(v, dtype=np.complex128).tobytes())
    return {"ints": hi.hexdigest(), "floats": hf.hexdigest(), "vecs": hv.hexdigest()}


def rng_digest():
    st = np.random.get_state()
    h = hashlib.sha256()
    h.update(np.asarray(st[1], dtype=np.uint32).tobytes())
    h.update(np.array([st[2]], dtype=np.int64).tobytes())
    py = random.getstate()
    h.update(np.array(py[1], dtype=np.uint64).tobytes())
    return h.hexdigest()


def full_rows(rows, limit=24):
    out = []
    for r in rows[:limit]:
        out.append({
            "id": int(r["id"]), "state": int(r["state"]), "stuck": int(r["stuck"]),
            "retries": int(r["retries"]), "resets": int(r["resets"]),
            "w": _f(r["w"]), "resid": _f(r["resid"]), "alpha": _c(r["alpha"]),
            "lam": _c(r["lam"] if r["lam"] is not None else 0j),
        })
    return out


def globals_record(energy, avg_resid, avg_stuck, n_distinct, stability, pref, strat):
    return {
        "energy": _f(energy), "avg_resid": _f(avg_resid), "avg_stuck": _f(avg_stuck),
        "n_distinct": int(n_distinct), "stability": stability, "pref": pref,
        "aggr": _f(strat["overall_psi_aggression_factor"]),
        "spawn": _f(strat["spawn_rate_multiplier"]),
        "thr": _f(strat["current_convergence_threshold"]),
        "max_retries": int(strat["max_psi_retries"]),
    }
