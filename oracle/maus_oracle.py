"""CPU oracle for the MAUS per-candidate hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a NumPy/SciPy *restatement* of the arithmetic performed by the
reference `Adaptive_Matrix_Solver_0.1.py` (cited below as AMS:line) on the path
named by BASELINE.json's north_star:

    SolutionCandidate.update_solution_step  ->  InverseIterateSolver.solve
    (+ the loop body of MAUS_Solver.evolve that drives it)

It is the checker for the HIP path.  Only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import it; the product package
(`adaptive_matrix_solver_amd/`) never does and fails loudly when the HIP
library is missing.

Parity pinning: the reference ships no tests or golden vectors, so
the oracle is pinned by fixtures captured from the reference itself, imported
unmodified in the build container (tests/golden/make_goldens.py, versions
recorded in every fixture).  tests/test_oracle_golden.py checks this module
against them bit-for-bit (same LAPACK/BLAS calls in the same order).

Style note: the reference is three mutable classes; this restatement is
deliberately functional (plain state records + free functions) so it can be
driven per candidate or over a whole population, and so the device path's
batched phases (rayleigh -> shifted solve -> relax/normalise -> residual) can
each be checked in isolation.

Both global RNG streams of the reference are consumed in the reference's order
(SURVEY appendix A): `np.random` (legacy MT19937) and Python's `random`.
"""
from __future__ import annotations

import random as _pyrandom
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import scipy.linalg as sla
import scipy.sparse.linalg as spla

# --------------------------------------------------------------------------
# constants (AMS:16-26)
# --------------------------------------------------------------------------
PSI_EPSILON_BASE = np.complex128(1e-20)      # AMS:16
ALPHA_INITIAL = np.complex128(0.01)          # AMS:17
MAX_PSI_ATTEMPTS = 25                        # AMS:18
MAX_STUCK_FOR_RETIREMENT = 8                 # AMS:19
MIN_WEIGHT_TO_SURVIVE = 1e-10                # AMS:20
VECTOR_SIMILARITY_TOL = 0.999                # AMS:21
LAMBDA_SIMILARITY_TOL = 1e-5                 # AMS:22
SIGMA_SIMILARITY_TOL_ABS = 1e-6              # AMS:23
SIGMA_SIMILARITY_TOL_REL = 1e-4              # AMS:24
CONVERGENCE_RESIDUAL_TOL = 1e-8              # AMS:25
MAX_STUCK_FOR_PRUNING = 4                    # AMS:26

# problem kinds (AMS:10-13) and candidate states (AMS:109-110) as plain ints
EIGENVALUE, SOLVE_LINEAR_SYSTEM, SVD = 1, 2, 3
EXPLORING, REFINING, STUCK, CONVERGED, RETIRED = 1, 2, 3, 4, 5

DIRECT, GMRES = "direct_solve", "iterative_gmres"


class SolveFailed(RuntimeError):
    """All psi attempts failed (AMS:104 raises RuntimeError)."""


# --------------------------------------------------------------------------
# L1: InverseIterateSolver.solve  (AMS:39-104)
# --------------------------------------------------------------------------
def psi_magnitude(base_psi, attempt: int, stuck: int):
    """AMS:44 -- psi = base * 10^(attempt/2) * 10^(stuck/3), evaluated left to right."""
    return base_psi * (10 ** (attempt / 2.0)) * (10 ** (stuck / 3.0))


def dense_regulariser(n: int, psi, dtype=np.complex128) -> np.ndarray:
    """AMS:49-50 -- psi*I + 0.15*psi*((U1-.5) + i(U2-.5)); U1 drawn before U2
    from the global legacy NumPy stream (4*n*n MT19937 words)."""
    u_re = np.random.rand(n, n)
    u_im = np.random.rand(n, n)
    pert = (u_re - 0.5 + 1j * (u_im - 0.5)) * psi * 0.15
    return psi * np.eye(n, dtype=dtype) + pert


def jacobi_inverse_diagonal(H: np.ndarray, stuck: int) -> Optional[np.ndarray]:
    """AMS:64-86 -- the Jacobi preconditioner is built only when stuck > 1, and
    only if every diagonal entry is finite after inversion and |d| > 1e-12.
    Returns the inverse diagonal (a vector); the reference materialises
    np.diag(inv) and multiplies by it, which is the same map up to the
    summation of exact zeros."""
    if not (stuck > 1 and H.shape[0] > 0):
        return None
    d = H.diagonal()
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
    if np.all(np.isfinite(inv)) and np.all(np.abs(d) > 1e-12):
        return inv
    return None


def gmres_scipy(H, b, x0, inv_diag, rtol=1e-8, maxiter=50):
    """The reference's call spla.gmres(H, b, x0=..., tol=1e-8, maxiter=50, M=...)
    (AMS:89) with the evident intent tol->rtol.  M is passed as the
    dense 2-D diagonal matrix exactly as the reference builds it (AMS:76)."""
    M = None if inv_diag is None else np.diag(inv_diag)
    return spla.gmres(H, b, x0=x0, rtol=rtol, maxiter=maxiter, M=M)


def gmres_restated(H, b, x0, inv_diag, rtol=1e-8, maxiter=50, restart=20):
    """Own restatement of SciPy 1.15's GMRES (scipy/sparse/linalg/_isolve/
    iterative.py:692-841): left-preconditioned, restart=min(20,n), modified
    Gram-Schmidt, Givens via the LAPACK zlartg convention, adaptive inner
    tolerance `ptol`, true-residual exit.  Returns (x, info, inner_iters,
    cycles).  This is the algorithm the device GMRES kernel implements; it is
    checked against scipy itself in tests/test_oracle_golden.py."""
    n = b.shape[0]
    b = np.asarray(b, dtype=np.complex128)
    x = np.array(x0, dtype=np.complex128, copy=True)

    def psolve(v):
        return v.copy() if inv_diag is None else inv_diag * v

    bnrm2 = np.linalg.norm(b)
    atol = max(0.0, rtol * float(bnrm2))            # _get_atol_rtol (iterative.py:19)
    if bnrm2 == 0:
        return b.copy(), 0, 0, 0
    eps = np.finfo(np.float64).eps
    restart = min(restart, n)
    Mb_nrm2 = np.linalg.norm(psolve(b))
    ptol_max_factor = 1.0
    ptol = Mb_nrm2 * min(ptol_max_factor, atol / bnrm2)
    presid = 0.0
    V = np.empty((restart + 1, n), dtype=np.complex128)
    Hh = np.zeros((restart, restart + 1), dtype=np.complex128)
    giv = np.zeros((restart, 2), dtype=np.complex128)
    inner = 0
    cycles = 0
    rnorm = np.inf
    for cycle in range(maxiter):
        if cycle == 0:
            r = b - H @ x if x.any() else b.copy()
            if np.linalg.norm(r) < atol:
                return x, 0, inner, cycles
        cycles += 1
        V[0] = psolve(r)
        tmp = np.linalg.norm(V[0])
        V[0] *= (1 / tmp)
        S = np.zeros(restart + 1, dtype=np.complex128)
        S[0] = tmp
        breakdown = False
        col = 0
        for col in range(restart):
            w = psolve(H @ V[col])
            h0 = np.linalg.norm(w)
            for k in range(col + 1):
                t = np.vdot(V[k], w)
                Hh[col, k] = t
                w -= t * V[k]
            h1 = np.linalg.norm(w)
            Hh[col, col + 1] = h1
            V[col + 1] = w
            if h1 <= eps * h0:
                Hh[col, col + 1] = 0
                breakdown = True
            else:
                V[col + 1] *= (1 / h1)
            for k in range(col):
                c, s = giv[k, 0], giv[k, 1]
                n0, n1 = Hh[col, k], Hh[col, k + 1]
                Hh[col, k], Hh[col, k + 1] = c * n0 + s * n1, -np.conj(s) * n0 + c * n1
            c, s, mag = zlartg(Hh[col, col], Hh[col, col + 1])
            giv[col] = (c, s)
            Hh[col, col], Hh[col, col + 1] = mag, 0
            t = -np.conjugate(s) * S[col]
            S[col], S[col + 1] = c * S[col], t
            presid = np.abs(t)
            inner += 1
            if presid <= ptol or breakdown:
                break
        if Hh[col, col] == 0:
            S[col] = 0
        y = np.array(S[: col + 1], dtype=np.complex128)
        for k in range(col, 0, -1):
            if y[k] != 0:
                y[k] /= Hh[k, k]
                t = y[k]
                y[:k] -= t * Hh[k, :k]
        if y[0] != 0:
            y[0] /= Hh[0, 0]
        x += y @ V[: col + 1]
        r = b - H @ x
        rnorm = np.linalg.norm(r)
        if rnorm <= atol:
            break
        elif breakdown:
            break
        elif presid <= ptol:
            ptol_max_factor = max(eps, 0.25 * ptol_max_factor)
        else:
            ptol_max_factor = min(1.0, 1.5 * ptol_max_factor)
        ptol = presid * min(ptol_max_factor, atol / rnorm)
    info = 0 if rnorm <= atol else maxiter
    return x, info, inner, cycles


def zlartg(f, g):
    """Complex plane rotation with real cosine, LAPACK 3.10+ zlartg semantics
    for unscaled (safe-range) inputs:  [c s; -conj(s) c] [f; g] = [r; 0]."""
    f = complex(f)
    g = complex(g)
    if g == 0:
        return 1.0, 0j, f
    if f == 0:
        d = abs(g)
        return 0.0, np.conj(g) / d, d
    f2 = f.real * f.real + f.imag * f.imag
    g2 = g.real * g.real + g.imag * g.imag
    h2 = f2 + g2
    # safe-range branch of LAPACK's la_lartg (f2 >= h2*safmin): d = sqrt(f2*h2)
    if f2 >= h2 * 2.2250738585072014e-308:
        c = np.sqrt(f2 / h2)
        r = f / c
        d = np.sqrt(f2 * h2) if (f2 > 1.4916681462400413e-154 and h2 < 6.703903964971299e+153) else None
        if d is not None:
            s = np.conj(g) * (f / d)
        else:
            s = np.conj(g) * (r / h2)
    else:
        d = np.sqrt(f2 * h2)
        c = f2 / d
        r = f / c if c >= 2.2250738585072014e-308 else f * (h2 / d)
        s = np.conj(g) * (f / d)
    return float(c), s, r


def inverse_iterate_solve(A_target: np.ndarray, rhs: np.ndarray, stuck: int, *,
                          n: int, base_psi, max_attempts: int,
                          preferred: str = DIRECT,
                          gmres_mode: str = "scipy-legacy",
                          trace: Optional[list] = None) -> Tuple[np.ndarray, int]:
    """AMS:39-104, dense branch only (sparse is out of scope).

    gmres_mode selects how the reference's `spla.gmres(..., tol=1e-8)` call
    (AMS:89) is honoured:
      "scipy-legacy": behave like the reference under SciPy >= 1.14, i.e. the
                      keyword is rejected with TypeError, which AMS:98 swallows;
      "rtol":         the evident intent, rtol=1e-8, through SciPy's gmres;
      "restated":     same, through gmres_restated() above.
    `trace`, if given, receives one dict per attempt (method, psi, outcome)."""
    fallback = GMRES if preferred == DIRECT else DIRECT       # AMS:36
    attempts = 0
    method = preferred
    while attempts < max_attempts:                              # AMS:43
        psi = psi_magnitude(base_psi, attempts, stuck)          # AMS:44
        H = A_target + dense_regulariser(n, psi, A_target.dtype)  # AMS:49-52
        rec = {"method": method, "attempt": attempts, "psi": psi}
        try:
            if method == DIRECT:
                x = sla.solve(H, rhs, assume_a="general")       # AMS:59
            elif method == GMRES:
                x0 = rhs if rhs.shape == H.shape[1:] else np.zeros_like(rhs)   # AMS:61
                inv_d = jacobi_inverse_diagonal(H, stuck)       # AMS:64-86
                rec["jacobi"] = inv_d is not None
                if gmres_mode == "scipy-legacy":
                    raise TypeError("gmres() got an unexpected keyword argument 'tol'")
                if gmres_mode == "rtol":
                    x, info = gmres_scipy(H, rhs, x0, inv_d)
                else:
                    x, info, inner, cyc = gmres_restated(H, rhs, x0, inv_d)
                    rec["inner"] = inner
                    rec["cycles"] = cyc
                rec["info"] = info
                if info != 0:                                   # AMS:90
                    raise np.linalg.LinAlgError("GMRES did not converge cleanly")
            else:
                raise ValueError("unknown method")
            if not np.all(np.isfinite(x)):                      # AMS:94-95
                raise ValueError("non-finite solution")
            rec["ok"] = True
            if trace is not None:
                trace.append(rec)
            return x, attempts                                  # AMS:97
        except (np.linalg.LinAlgError, ValueError, TypeError):  # AMS:98
            rec["ok"] = False
            if trace is not None:
                trace.append(rec)
            if method == preferred and preferred != fallback and attempts == 0:   # AMS:99
                method = fallback
                attempts = 0
                continue
            attempts += 1                                       # AMS:103
    raise SolveFailed("all psi attempts failed")                # AMS:104


# --------------------------------------------------------------------------
# L2: candidate state + one step  (AMS:107-337)
# --------------------------------------------------------------------------
@dataclass
class Cand:
    """Per-candidate record (AMS:112-127).  `v` doubles as v_k / right_v_k."""
    cid: int
    kind: int
    n: int
    rows: int
    cols: int
    A_pm: Any                       # problem_matrix captured at construction (AMS:118, F9)
    lam: Any = None
    v: Optional[np.ndarray] = None
    x: Optional[np.ndarray] = None
    sigma: Any = None
    u: Optional[np.ndarray] = None
    state: int = EXPLORING
    w: float = 0.01
    resid: float = float("inf")
    prev_resid: float = float("inf")
    alpha: Any = ALPHA_INITIAL
    stuck: int = 0
    retries: int = 0
    resets: int = 0
    resid_hist: List[float] = field(default_factory=list)
    param_hist: List[tuple] = field(default_factory=list)

    def params(self):
        """AMS:333-337."""
        if self.kind == EIGENVALUE:
            return (self.lam, self.v)
        if self.kind == SOLVE_LINEAR_SYSTEM:
            return (self.x,)
        return (self.sigma, self.u, self.v)


class IdCounter:
    """The class-global id counter (AMS:108, reset quirk AMS:368 / SURVEY F12)."""
    value = 0


def _rand_c(n):
    return (np.random.rand(n) + 1j * np.random.rand(n)).astype(np.complex128)   # AMS:130


def _unit_rand(n):
    """AMS:131 -- v/||v|| (norm evaluated twice in the reference; value identical)."""
    v = _rand_c(n)
    nv = np.linalg.norm(v)
    if nv > 1e-10:
        return v / np.linalg.norm(v)
    w = _rand_c(n)
    return w / np.linalg.norm(_rand_c(n))


def random_init(c: Cand) -> None:
    """AMS:129-143 (RNG event E1 of SURVEY appendix A)."""
    if c.kind == EIGENVALUE:
        c.v = _unit_rand(c.n)
        c.lam = (_pyrandom.random() * 5 - 2.5 + 1j * (_pyrandom.random() * 5 - 2.5))
    elif c.kind == SOLVE_LINEAR_SYSTEM:
        c.x = _unit_rand(c.n) * _pyrandom.uniform(0.1, 10.0)
    else:
        c.u = _unit_rand(c.rows)
        c.v = _unit_rand(c.cols)
        c.sigma = 1.0
    c.param_hist.append(c.params())
    c.resid_hist.append(c.resid)


def new_candidate(A_pm, kind: int, n_diag: int, *, weight=0.01) -> Cand:
    """AMS:112-127.  Seeds passed by the spawner are overwritten by the random
    init, so they are not parameters here."""
    rows, cols = A_pm.shape
    c = Cand(cid=IdCounter.value, kind=kind, n=n_diag, rows=rows, cols=cols, A_pm=A_pm, w=weight)
    IdCounter.value += 1
    random_init(c)
    return c


def rayleigh(A, v):
    """AMS:264-268."""
    den = np.vdot(v, v)
    if np.abs(den) < 1e-12:
        return complex(0.0, 0.0)
    return np.vdot(v, A @ v) / den


def hermitian_shortcut(c: Cand, A) -> bool:
    """AMS:155-181 dense branch: full eigh per candidate, best overlap wins."""
    try:
        evals, evecs = sla.eigh(A)
        if c.v is not None and evecs.shape[1] > 0:
            scores = np.abs(c.v.conj().T @ evecs)                 # AMS:165
            j = int(np.argmax(scores))                            # AMS:169
            c.lam = evals[j]
            c.v = evecs[:, j]
            c.v /= np.linalg.norm(c.v)                            # AMS:173 (writes through the view)
            c.resid = np.linalg.norm(A @ c.v - c.lam * c.v)       # AMS:175
            c.state = CONVERGED
            c.stuck = 0
            c.retries = 0
            c.w = 1.0
            return True
    except Exception:
        pass
    return False


def svd_power_step(c: Cand, A, strat) -> None:
    """AMS:227-255."""
    try:
        if np.linalg.norm(c.v) < 1e-10:
            c.v = (np.random.rand(c.cols) + 1j * np.random.rand(c.cols))
            c.v /= np.linalg.norm(c.v)
            c.stuck += 1
            c.resets += 1
            raise ValueError("right vector collapsed")
        t = A @ c.v
        c.sigma = np.linalg.norm(t)
        c.u = t / (c.sigma if c.sigma > 1e-10 else 1.0)
        if np.linalg.norm(c.u) < 1e-10:
            c.u = (np.random.rand(c.rows) + 1j * np.random.rand(c.rows))
            c.u /= np.linalg.norm(c.u)
            c.stuck += 1
            c.resets += 1
            raise ValueError("left vector collapsed")
        s = A.conj().T @ c.u
        c.sigma = max(c.sigma, np.linalg.norm(s))
        c.v = s / (np.linalg.norm(s) if np.linalg.norm(s) > 1e-10 else 1.0)
        if c.sigma < SIGMA_SIMILARITY_TOL_ABS / 100:
            c.resid = strat.get("current_convergence_threshold", 1e-6) * 0.1
            c.state = CONVERGED
            c.stuck = 0
            if np.linalg.norm(c.u) < 1e-10:
                c.u = np.ones(c.rows, dtype=np.complex128) / np.sqrt(c.rows)
            if np.linalg.norm(c.v) < 1e-10:
                c.v = np.ones(c.cols, dtype=np.complex128) / np.sqrt(c.cols)
        else:
            c.stuck = max(0, c.stuck - 1)
    except (RuntimeError, ValueError, np.linalg.LinAlgError):
        c.stuck += 1
        c.w *= 0.001
        c.alpha *= 0.5
        c.state = STUCK
        if c.stuck >= MAX_STUCK_FOR_RETIREMENT:
            c.state = RETIRED
        c.u = (np.random.rand(c.rows) + 1j * np.random.rand(c.rows)) / np.sqrt(c.rows)
        c.v = (np.random.rand(c.cols) + 1j * np.random.rand(c.cols)) / np.sqrt(c.cols)
        c.sigma = 1.0


def residual_of(c: Cand, b) -> float:
    """AMS:295-301 -- always against the construction-time matrix (F9)."""
    A = c.A_pm
    if c.kind == EIGENVALUE:
        return np.linalg.norm(A @ c.v - c.lam * c.v) if c.v is not None else float("inf")
    if c.kind == SOLVE_LINEAR_SYSTEM:
        return np.linalg.norm(A @ c.x - b) if (c.x is not None and b is not None) else float("inf")
    if c.v is not None and c.u is not None:
        return (np.linalg.norm(A @ c.v - c.sigma * c.u)
                + np.linalg.norm(A.conj().T @ c.u - c.sigma * c.v))
    return float("inf")


def adapt_alpha_state(c: Cand) -> None:
    """AMS:306-316."""
    if c.prev_resid > 1e-10:
        if c.resid < c.prev_resid * 0.9:
            c.alpha = min(c.alpha * 1.1, 1.0)
            if c.state != CONVERGED:
                c.state = REFINING
        elif c.resid > c.prev_resid * 1.5 and c.prev_resid > 1e-5:
            c.alpha = max(c.alpha * 0.5, 1e-6)
            if c.state != CONVERGED:
                c.state = STUCK
        else:
            c.alpha = max(c.alpha * 0.95, 1e-6)
            if c.state not in (CONVERGED, STUCK, RETIRED):
                c.state = EXPLORING


def params_finite(c: Cand) -> bool:
    """AMS:319-327."""
    for p in c.params():
        if p is None:
            return False
        if isinstance(p, np.ndarray):
            if not np.all(np.isfinite(p)):
                return False
        elif not np.isfinite(p):
            return False
    return True


def candidate_step(c: Cand, A, b, strat: Dict, know: Dict, *, gmres_mode="scipy-legacy",
                   trace: Optional[list] = None) -> None:
    """One update_solution_step (AMS:145-331), dense matrices."""
    c.prev_resid = c.resid                                        # AMS:147
    aggr = strat.get("overall_psi_aggression_factor", 1.0)
    max_retries = strat.get("max_psi_retries", MAX_PSI_ATTEMPTS)
    pref = know.get("local_solver_preference", DIRECT)

    if c.kind == EIGENVALUE and know.get("is_hermitian", False):   # AMS:155
        if hermitian_shortcut(c, A):
            c.param_hist.append(c.params())
            c.resid_hist.append(c.resid)
            return                                                 # AMS:218-221

    base_psi = PSI_EPSILON_BASE * aggr                             # AMS:224

    if c.kind == SVD:
        svd_power_step(c, A, strat)
    else:
        if c.kind == EIGENVALUE:
            if np.linalg.norm(c.v) < 1e-10:                        # AMS:259-263 (E2)
                c.v = (np.random.rand(c.n) + 1j * np.random.rand(c.n))
                c.v /= np.linalg.norm(c.v)
                c.stuck += 1
                c.resets += 1
            c.lam = rayleigh(A, c.v)                               # AMS:264-268
            target = A - c.lam * np.eye(c.n, dtype=A.dtype)        # AMS:270
            rhs = c.v
        else:
            target = A                                             # AMS:274-276
            rhs = b
        try:
            wvec, c.retries = inverse_iterate_solve(                # AMS:278
                target, rhs, c.stuck, n=c.n, base_psi=base_psi, max_attempts=max_retries,
                preferred=pref, gmres_mode=gmres_mode, trace=trace)
            if c.kind == EIGENVALUE:
                c.v = (1.0 - c.alpha) * c.v + c.alpha * wvec       # AMS:280
                nv = np.linalg.norm(c.v)
                if nv > 1e-10:
                    c.v /= nv
                else:                                              # AMS:283 (E4)
                    c.v = (np.random.rand(c.n) + 1j * np.random.rand(c.n)) / np.sqrt(c.n)
            else:
                c.x = (1.0 - c.alpha) * c.x + c.alpha * wvec       # AMS:285
            c.stuck = max(0, c.stuck - 1)                          # AMS:286
        except (RuntimeError, ValueError):                          # AMS:287-293
            c.stuck += 1
            c.w *= 0.001
            c.alpha = max(c.alpha * 0.5, 1e-6)
            if c.stuck >= MAX_STUCK_FOR_RETIREMENT:
                c.state = RETIRED
                c.resets += 1
            else:
                c.state = STUCK
                random_init(c)                                     # E5

    c.resid = residual_of(c, b)                                    # AMS:295-301
    c.param_hist.append(c.params())                                # AMS:303-304
    c.resid_hist.append(c.resid)
    adapt_alpha_state(c)                                           # AMS:306-316
    thr = strat.get("current_convergence_threshold", CONVERGENCE_RESIDUAL_TOL)
    if c.resid < thr and params_finite(c):                         # AMS:329-331
        c.state = CONVERGED
        c.w = 1.0
        c.stuck = 0
        c.alpha = 0.0


# --------------------------------------------------------------------------
# L3: population manager pieces that the loop body runs  (AMS:340-549, 572-577)
# --------------------------------------------------------------------------
@dataclass
class Pop:
    M: np.ndarray
    kind: int
    b: Optional[np.ndarray]
    n_rows: int
    n_cols: int
    n: int
    diag: Dict
    know: Dict
    strat: Dict
    cands: List[Cand]
    energy: float = 1.0
    avg_resid: float = 1.0
    avg_stuck: float = 0.0
    n_distinct: int = 0
    converged: List[tuple] = field(default_factory=list)


def diagnose_matrix(M: np.ndarray) -> Dict:
    """AMS:374-404, ndarray branch."""
    d = {"is_hermitian": False, "is_complex_symmetric": False, "is_sparse_init": False,
         "condition_number": np.inf, "is_singular": False}
    d["is_sparse_init"] = (np.count_nonzero(M) / M.size) < 0.25 if M.size > 0 else False
    if M.ndim == 2 and M.shape[0] == M.shape[1]:
        if np.allclose(M, M.conj().T):
            d["is_hermitian"] = True
        if np.allclose(M, M.T):
            d["is_complex_symmetric"] = True
    cond, sing = np.inf, False
    if (not d["is_sparse_init"]) and M.ndim == 2 and M.shape[0] == M.shape[1] and M.size > 0:
        try:
            cond = np.linalg.cond(M)
            if np.isinf(cond) or cond > 1e15:
                sing = True
        except np.linalg.LinAlgError:
            cond, sing = np.inf, True
    d["condition_number"] = cond
    d["is_singular"] = sing
    return d


def initial_strategy(pop: Pop) -> None:
    """AMS:406-422."""
    s, k = pop.strat, pop.know
    cond = pop.diag["condition_number"]
    if cond > 1e12:
        k["numerical_stability_state"] = "Critical"
        s["overall_psi_aggression_factor"] = 50.0
        s["max_psi_retries"] = MAX_PSI_ATTEMPTS * 2
        s["current_convergence_threshold"] = 1e-2
        k["local_solver_preference"] = GMRES
    elif cond > 1e6:
        k["numerical_stability_state"] = "Fragile"
        s["overall_psi_aggression_factor"] = 10.0
        k["local_solver_preference"] = GMRES
        s["current_convergence_threshold"] = 1e-4
    else:
        k["numerical_stability_state"] = "Stable"
        k["local_solver_preference"] = DIRECT
        s["current_convergence_threshold"] = s["convergence_tolerance"]
    if pop.kind == SOLVE_LINEAR_SYSTEM and pop.diag.get("is_singular", False):
        k["true_matrix_is_singular"] = True
        k["local_solver_preference"] = GMRES
        s["overall_psi_aggression_factor"] = max(s["overall_psi_aggression_factor"], 20.0)
    if pop.kind == SVD:
        if k["numerical_stability_state"] == "Stable":
            s["overall_psi_aggression_factor"] = max(s["overall_psi_aggression_factor"], 2.0)
        s["current_convergence_threshold"] = max(1e-5, s["convergence_tolerance"])


def new_population(M, kind: int, b=None, n_cands: Optional[int] = None, tol: float = 1e-8) -> Pop:
    """AMS:341-372, dense ndarray input."""
    M = M.astype(np.complex128)
    n_rows, n_cols = M.shape
    n = n_rows
    b = b.astype(np.complex128) if b is not None else None
    diag = diagnose_matrix(M)
    know = {"matrix_type": "Dense", "spectrum_hint": "Unknown", "numerical_stability_state": "Stable",
            "local_solver_preference": DIRECT, "effective_rank_SVD": min(n_rows, n_cols),
            "true_matrix_is_singular": diag["is_singular"], "is_sparse_problem": diag["is_sparse_init"],
            "is_hermitian": diag.get("is_hermitian", False),
            "is_complex_symmetric": diag.get("is_complex_symmetric", False)}
    strat = {"overall_psi_aggression_factor": 1.0, "max_psi_retries": MAX_PSI_ATTEMPTS,
             "min_survival_weight": MIN_WEIGHT_TO_SURVIVE, "spawn_rate_multiplier": 1.0,
             "convergence_tolerance": tol, "current_convergence_threshold": tol}
    pop = Pop(M=M, kind=kind, b=b, n_rows=n_rows, n_cols=n_cols, n=n, diag=diag, know=know,
              strat=strat, cands=[])
    initial_strategy(pop)
    k = n_cands if n_cands is not None else n * 3
    if kind == SVD:
        k = max(k, min(n_rows, n_cols) * 3)
    pop.cands = [new_candidate(pop.M, kind, n) for _ in range(k)]
    IdCounter.value = k                                            # AMS:368
    return pop


def _same_eig(lam_a, v_a, lam_s, v_s) -> bool:
    tol = LAMBDA_SIMILARITY_TOL + np.abs(lam_s) * 1e-6
    return bool(np.abs(lam_a - lam_s) < tol and np.abs(np.vdot(v_a, v_s)) > VECTOR_SIMILARITY_TOL)


def update_diagnostics(pop: Pop) -> None:
    """AMS:424-475."""
    total = len(pop.cands)
    sum_res = 0.0
    sum_stuck = 0
    n_conv = 0
    pop.n_distinct = 0
    pop.converged = []
    sigmas = []
    thr = pop.strat["current_convergence_threshold"]
    for c in pop.cands:
        if c.state == CONVERGED:
            n_conv += 1
            tup = c.params()
            distinct = True
            if tup is None or any(p is None for p in tup):
                continue
            if pop.kind == EIGENVALUE:
                for s in pop.converged:
                    if _same_eig(tup[0], tup[1], s[0], s[1]):
                        distinct = False
                        break
            elif pop.kind == SOLVE_LINEAR_SYSTEM:
                if (len(pop.converged) > 0 and
                        np.linalg.norm(tup[0] - pop.converged[0][0]) < pop.strat["convergence_tolerance"] * 100):
                    distinct = False
            else:
                max_s = max((q.sigma.real for q in pop.cands
                             if q.sigma is not None and q.sigma.real > 0), default=1.0)
                if tup[0].real / max_s < SIGMA_SIMILARITY_TOL_REL:
                    distinct = False
                if distinct:
                    for s in pop.converged:
                        if (np.abs(tup[0] - s[0]) < max(SIGMA_SIMILARITY_TOL_ABS, s[0] * SIGMA_SIMILARITY_TOL_REL)
                                and np.abs(np.vdot(tup[1], s[1])) > VECTOR_SIMILARITY_TOL
                                and np.abs(np.vdot(tup[2], s[2])) > VECTOR_SIMILARITY_TOL):
                            distinct = False
                            break
                sigmas.append(tup[0].real)
            if distinct:
                pop.converged.append(tup)
                pop.n_distinct += 1
        if c.state not in (CONVERGED, RETIRED):
            sum_res += c.resid if np.isfinite(c.resid) else thr * 100
            sum_stuck += c.stuck
    den = max(1, total - n_conv)
    pop.avg_resid = sum_res / den
    pop.avg_stuck = sum_stuck / den
    norm_res = pop.avg_resid / (thr * 10)
    norm_stuck = pop.avg_stuck / (MAX_STUCK_FOR_RETIREMENT * 2)
    target = pop.n
    if pop.kind == SOLVE_LINEAR_SYSTEM:
        target = 1
    elif pop.kind == SVD:
        if len(sigmas) > 1:
            ss = sorted([s for s in sigmas if s > SIGMA_SIMILARITY_TOL_ABS], reverse=True)
            if ss:
                rank = sum(1 for s in ss if s / ss[0] > SIGMA_SIMILARITY_TOL_REL)
                pop.know["effective_rank_SVD"] = min(rank if rank > 0 else 1, min(pop.n_rows, pop.n_cols),
                                                     max(1, pop.know.get("effective_rank_SVD", 1)))
        target = pop.know.get("effective_rank_SVD", min(pop.n_rows, pop.n_cols))
    missing = (target - pop.n_distinct) / max(1, target)
    pop.energy = max(0.0, min(1.0, norm_res * 0.4 + norm_stuck * 0.3 + missing * 0.3))
    if pop.avg_stuck > MAX_STUCK_FOR_RETIREMENT * 0.5:
        pop.know["numerical_stability_state"] = "Critical"
    elif pop.avg_stuck > MAX_STUCK_FOR_PRUNING * 0.5:
        pop.know["numerical_stability_state"] = "Fragile"
    else:
        pop.know["numerical_stability_state"] = "Stable"


def adjust_strategy(pop: Pop) -> None:
    """AMS:477-501."""
    s, k = pop.strat, pop.know
    stab = k["numerical_stability_state"]
    tol = s["convergence_tolerance"]
    if pop.energy > 0.6 and stab == "Critical":
        k["local_solver_preference"] = GMRES
        s["overall_psi_aggression_factor"] = min(200.0, s["overall_psi_aggression_factor"] * 1.1)
        s["spawn_rate_multiplier"] = min(10.0, s["spawn_rate_multiplier"] * 1.2)
        s["current_convergence_threshold"] = max(tol * 50, s["current_convergence_threshold"] * 1.05)
    elif pop.energy > 0.4 and stab == "Fragile":
        k["local_solver_preference"] = GMRES
        s["overall_psi_aggression_factor"] = min(50.0, s["overall_psi_aggression_factor"] * 1.05)
        s["spawn_rate_multiplier"] = min(5.0, s["spawn_rate_multiplier"] * 1.1)
        s["current_convergence_threshold"] = max(tol * 5, s["current_convergence_threshold"] * 1.02)
    elif pop.energy < 0.2 and stab == "Stable":
        k["local_solver_preference"] = DIRECT
        s["overall_psi_aggression_factor"] = max(1.0, s["overall_psi_aggression_factor"] * 0.9)
        s["spawn_rate_multiplier"] = max(0.01, s["spawn_rate_multiplier"] * 0.9)
        s["current_convergence_threshold"] = max(tol, s["current_convergence_threshold"] * 0.9)
    s["overall_psi_aggression_factor"] = max(1.0, min(200.0, s["overall_psi_aggression_factor"]))
    s["spawn_rate_multiplier"] = max(0.01, min(10.0, s["spawn_rate_multiplier"]))
    s["current_convergence_threshold"] = max(tol, min(1.0, s["current_convergence_threshold"]))


def manage_candidates(pop: Pop) -> None:
    """AMS:504-549."""
    survivors: List[Cand] = []
    order = sorted(pop.cands, key=lambda c: (-c.w, c.resid if np.isfinite(c.resid) else float("inf")))
    tol = pop.strat["convergence_tolerance"]
    for c in order:
        redundant = False
        if c.state == CONVERGED:
            for s in survivors:
                if s.state != CONVERGED:
                    continue
                tc, ts = c.params(), s.params()
                if tc is None or ts is None or any(p is None for p in tc) or any(p is None for p in ts):
                    continue
                if pop.kind == EIGENVALUE:
                    if _same_eig(tc[0], tc[1], ts[0], ts[1]):
                        redundant = True
                        break
                elif pop.kind == SOLVE_LINEAR_SYSTEM:
                    if np.linalg.norm(tc[0] - ts[0]) < tol * 10:
                        redundant = True
                        break
                else:
                    if ts[0].real < SIGMA_SIMILARITY_TOL_ABS / 100:
                        redundant = False
                    elif (np.abs(tc[0] - ts[0]) < max(SIGMA_SIMILARITY_TOL_ABS, ts[0] * SIGMA_SIMILARITY_TOL_REL)
                          and np.abs(np.vdot(tc[1], ts[1])) > VECTOR_SIMILARITY_TOL
                          and np.abs(np.vdot(tc[2], ts[2])) > VECTOR_SIMILARITY_TOL):
                        redundant = True
                        break
        if redundant:
            c.state = RETIRED
        elif c.state == RETIRED:
            pass
        elif ((c.w < pop.strat["min_survival_weight"] and c.state != CONVERGED)
              or (c.stuck >= MAX_STUCK_FOR_RETIREMENT and c.state != CONVERGED)):
            c.state = RETIRED
        else:
            survivors.append(c)
    pop.cands = survivors
    target = pop.n
    if pop.kind == SOLVE_LINEAR_SYSTEM:
        target = 1
    elif pop.kind == SVD:
        target = pop.know.get("effective_rank_SVD", min(pop.n_rows, pop.n_cols))
    base = max(5, int(pop.n * 1.5 if pop.kind == SOLVE_LINEAR_SYSTEM else pop.n * 2))
    if pop.kind == SVD:
        base = max(base, int(target * 2.5))
    k = max(0, base - len(pop.cands)) + max(0, target - pop.n_distinct)
    k = min(int(k * pop.strat["spawn_rate_multiplier"]), pop.n * 2, 15)
    for _ in range(max(0, k)):
        if pop.n_distinct > 0 and pop.energy < 0.8 and pop.converged:            # AMS:539 (E7)
            base_sol = _pyrandom.choice(pop.converged)
            if base_sol is None or any(p is None for p in base_sol):
                continue
            if pop.kind == EIGENVALUE:
                # the seeded values are discarded by the constructor (F8) but the draws happen
                _ = (_pyrandom.random() * 0.1 - 0.05 + 1j * (_pyrandom.random() * 0.1 - 0.05))
                pert = (np.random.rand(pop.n) - 0.5 + 1j * (np.random.rand(pop.n) - 0.5)) * (0.1 + pop.energy)
                new_v = base_sol[1] + pert
                if not (np.linalg.norm(new_v) > 1e-9):
                    _ = (np.random.rand(pop.n) + 1j * np.random.rand(pop.n))
        c = new_candidate(pop.M, pop.kind, pop.n, weight=0.01)
        c.alpha = ALPHA_INITIAL * (1 + pop.strat["overall_psi_aggression_factor"] / 10.0)   # AMS:548
        pop.cands.append(c)


def loop_body(pop: Pop, *, gmres_mode="scipy-legacy") -> int:
    """One iteration of the evolve loop (AMS:573-577).  Returns the number of
    candidate steps executed (the metric's unit)."""
    update_diagnostics(pop)
    adjust_strategy(pop)
    steps = 0
    for c in pop.cands:
        if c.state not in (CONVERGED, RETIRED):
            candidate_step(c, pop.M, pop.b, pop.strat, pop.know, gmres_mode=gmres_mode)
            steps += 1
    manage_candidates(pop)
    return steps


def bookkeeping(pop: Pop) -> List[tuple]:
    """The integer bookkeeping tuple compared bit-exactly between paths."""
    return [(c.cid, c.state, c.stuck, c.retries, c.resets) for c in pop.cands]


def seed_all(seed: int) -> None:
    """Harness obligation: both streams + the id counter."""
    np.random.seed(seed)
    _pyrandom.seed(seed)
    IdCounter.value = 0
