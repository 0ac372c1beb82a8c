"""Host-side mirror of the reference API for the MAUS hot path.

Same names, arguments, attributes and error behaviour as the reference's
`InverseIterateSolver` (AMS:30-104), `SolutionCandidate` (AMS:107-337) and `MAUS_Solver`
(AMS:340-608) -- AMS = Kier73/Adaptive-Matrix-Solver `Adaptive_Matrix_Solver_0.1.py` --
so user code switches by changing the import.  All matrix-sized arithmetic of the
candidate step runs in HIP kernels behind `libmaus_hip.so` (engine.py / _cabi.py); the
orchestration the reference keeps in Python (candidate spawn/retire, Psi aggression,
alpha adaptation, landscape-energy bookkeeping, both RNG streams) stays here, in the
reference's order, so bookkeeping is reproduced exactly.

Documented deviations:
  F1  evolve() defines the `target_sols_final` the reference forgot (= target_sols_disp).
  F2  `gmres_compat='rtol'` (default) honours the evident intent of `tol=1e-8`;
      `'scipy-legacy'` reproduces SciPy>=1.14 behaviour (TypeError swallowed -> LU).
  F5  the Hermitian eigendecomposition is computed once per matrix, not once per candidate.
  Sparse inputs are out of scope (dense BASELINE configs only) and raise NotImplementedError.
"""
from __future__ import annotations

import math
import random
from enum import Enum

import numpy as np

from . import _cabi
from ._cabi import POP_U, POP_X
from .candstore import C_NP as _C_NP, C_PY as _C_PY, EXACT as _EXACT, F_NP as _F_NP, F_PY as _F_PY, HREF as _HREF, MISSING as _MISSING
from .engine import DIRECT, GMRES, DeviceEngine, _advance_numpy_stream


class ProblemType(Enum):                     # AMS:10-13
    EIGENVALUE = 1
    SOLVE_LINEAR_SYSTEM = 2
    SVD = 3


# AMS:16-26
GLOBAL_DEFAULT_PSI_EPSILON_BASE = np.complex128(1e-20)
GLOBAL_DEFAULT_ALPHA_V_INITIAL = np.complex128(0.01)
REFERENCE_LAZY_MIN = 512          # evolve(): general eigenvalues of larger matrices are computed when true_solution is first read
_DEFERRED = object()


def _classify_real(v):
    """(numeric mirror, kind) of a value assigned to a real-valued candidate attribute (candstore.py)."""
    t = type(v)
    if t is np.float64:
        return v, _F_NP
    if t is float:
        return v, _F_PY
    try:
        return float(v), _EXACT
    except (TypeError, ValueError):
        return np.nan, _EXACT
GLOBAL_MAX_PSI_ATTEMPTS = 25
GLOBAL_MAX_STUCK_FOR_RETIREMENT = 8
GLOBAL_MIN_WEIGHT_TO_SURVIVE_PRUNE = 1e-10
GLOBAL_VECTOR_SIMILARITY_TOL = 0.999
GLOBAL_LAMBDA_SIMILARITY_TOL = 1e-5
GLOBAL_SIGMA_SIMILARITY_TOL_ABS = 1e-6
GLOBAL_SIGMA_SIMILARITY_TOL_REL = 1e-4
GLOBAL_CONVERGENCE_RESIDUAL_TOL = 1e-8
GLOBAL_MAX_STUCK_FOR_PRUNING = 4


def _allclose_to_transpose(M, conj: bool, rows_per_block: int = 256) -> bool:
    """np.allclose(M, M.conj().T) / np.allclose(M, M.T) evaluated block row by block row.  allclose is a conjunction of
    one elementwise predicate, so the answer is the same; a matrix that is not (conjugate-)symmetric is rejected at
    its first block instead of after a strided pass over all of it (3 s per check at n=4096)."""
    n = M.shape[0]
    for i in range(0, n, rows_per_block):
        other = M[:, i:i + rows_per_block].T
        if conj:
            other = other.conj()
        if not np.allclose(M[i:i + rows_per_block, :], other):
            return False
    return True


def _is_sparse(M) -> bool:
    try:
        import scipy.sparse as sp
        return sp.issparse(M)
    except Exception:
        return False


# ==========================================================================================
# InverseIterateSolver (AMS:30-104)
# ==========================================================================================
class InverseIterateSolver:
    """Psi-escalating regularised solve of (A_target + psi*I + pert) x = b on the GPU."""

    _engine = None          # private context: every call uploads its own A_target

    def __init__(self, N, base_psi_epsilon, max_attempts, preferred_method="direct_solve", is_sparse=False,
                 gmres_compat="rtol", pert_mode="uniform"):
        self.N = N
        self.base_psi_epsilon = base_psi_epsilon
        self.max_attempts = max_attempts
        self.preferred_method = preferred_method
        self.fallback_method = "iterative_gmres" if preferred_method == "direct_solve" else "direct_solve"
        self.is_sparse = is_sparse
        self.gmres_compat = gmres_compat
        self.pert_mode = pert_mode
        self.last_trace = []

    @classmethod
    def _ctx(cls):
        if cls._engine is None:
            cls._engine = _cabi.Context(0)
        return cls._engine

    def solve(self, A_target, b_rhs, candidate_stuck_counter):
        if self.is_sparse or _is_sparse(A_target):
            raise NotImplementedError("sparse problems are outside the MI355X hot path (dense only)")
        ctx = self._ctx()
        n = self.N
        A_target = np.ascontiguousarray(A_target, dtype=np.complex128)
        ctx.set_matrix(A_target)
        ctx.set_rhs(np.ascontiguousarray(b_rhs, dtype=np.complex128))
        ctx.pop_reserve(1)
        ctx.pop_put(POP_X, [0], np.ascontiguousarray(b_rhs, dtype=np.complex128))     # x0 = b (AMS:61)
        uniform = self.pert_mode == "uniform"
        num_psi_attempts = 0
        method = self.preferred_method
        self.last_trace = []
        zero = np.zeros(1, dtype=np.complex128)
        while num_psi_attempts < self.max_attempts:                                    # AMS:43
            psi = self.base_psi_epsilon * (10 ** (num_psi_attempts / 2.0)) * (10 ** (candidate_stuck_counter / 3.0))
            pert_data = None
            pmode = _cabi.PERT_UNIFORM if uniform else (_cabi.PERT_MT19937 if self.pert_mode == "mt19937" else _cabi.PERT_NONE)
            if uniform:
                pert_data = np.empty((1, 2, n, n))
                pert_data[0, 0] = np.random.rand(n, n)                                 # AMS:49
                pert_data[0, 1] = np.random.rand(n, n)
            else:
                if pmode == _cabi.PERT_MT19937:
                    pert_data = (np.random.get_state(), 4 * n * n, 0, np.zeros(1, dtype=np.int32))
                _advance_numpy_stream(4 * n * n)
            ps = np.array([complex(psi).real])
            rec = {"method": method, "attempt": num_psi_attempts, "psi": psi}
            try:
                if method == "direct_solve":
                    st = ctx.shifted_lu_solve([0], zero, ps, rhs_mode=1, pert_mode=pmode, pert_data=pert_data)[0]
                    if st > 0:
                        raise np.linalg.LinAlgError("Matrix is singular.")
                    if st == -1:
                        raise ValueError("array must not contain infs or NaNs")
                    if st == -2:
                        raise ValueError("Solution vector not finite after solve.")
                elif method == "iterative_gmres":
                    if self.gmres_compat == "scipy-legacy":
                        raise TypeError("gmres() got an unexpected keyword argument 'tol'")
                    typ = float(np.linalg.norm(A_target)) / np.sqrt(max(1, A_target.size))
                    if pmode != _cabi.PERT_NONE and 0.075 * abs(ps[0]) >= 2.0 ** -53 * typ:
                        # escalated psi: the random term of AMS:49-50 is no longer below the rounding of a matvec ->
                        # GMRES against the materialised H_solve (engine.DeviceEngine.pert_matters states the rule)
                        want = np.array([1 if (candidate_stuck_counter > 1 and n > 0) else 0], dtype=np.int32)
                        info, inner, status, jac = ctx.gmres_pert([0], zero, ps, 1, want, pmode, pert_data)
                        use_j = jac.astype(np.int32)
                        rec["jacobi"], rec["dense"] = bool(jac[0]), True
                    else:
                        use_j = np.zeros(1, dtype=np.int32)
                        if candidate_stuck_counter > 1 and n > 0 and ctx.jacobi_check(zero, ps)[0]:     # AMS:65-72
                            use_j[0] = 1
                        rec["jacobi"] = bool(use_j[0])
                        info, inner, status = ctx.gmres([0], zero, ps, 1, use_j)
                    rec["info"], rec["inner"] = int(info[0]), int(inner[0])
                    if status[0] == -1:
                        raise ValueError("array must not contain infs or NaNs")
                    if info[0] != 0:
                        raise np.linalg.LinAlgError(f"GMRES did not converge cleanly (info={info[0]}). "
                                                    f"Preconditioned: {'Yes' if use_j[0] else 'No'}")
                    if status[0] == -2:
                        raise ValueError("Solution vector not finite after solve.")
                else:
                    raise ValueError(f"Unknown solver method: {method}")
                rec["ok"] = True
                self.last_trace.append(rec)
                x = ctx.pop_get(_cabi.POP_W, [0], n)[0]
                return x, num_psi_attempts                                             # AMS:97
            except (np.linalg.LinAlgError, ValueError, TypeError):
                rec["ok"] = False
                self.last_trace.append(rec)
                if method == self.preferred_method and self.preferred_method != self.fallback_method and num_psi_attempts == 0:
                    method = self.fallback_method
                    num_psi_attempts = 0
                    continue
                num_psi_attempts += 1
        raise RuntimeError(f"InverseIterateSolver failed all {self.max_attempts} attempts for "
                           f"{self.preferred_method} and {self.fallback_method}.")


class _HistRef:
    """One recorded iterate whose vectors live in the context's history store (device, spilled to host when old).
    The store is dropped when the context is rebound to a matrix with another vector length (engine reuse) or cleared;
    the reference records the store's generation so that a stale index is refused instead of resolving to the rows
    of whoever appended next."""
    __slots__ = ("ctx", "scalar", "rows", "gen")

    def __init__(self, ctx, scalar, rows, gen=None):
        self.ctx, self.scalar, self.rows = ctx, scalar, rows          # rows: ((history index, length), ...)
        if gen is None:
            gen = ctx.hist_generation() if hasattr(ctx, "hist_generation") else 0
        self.gen = gen

    def resolve(self):
        if hasattr(self.ctx, "hist_generation") and self.ctx.hist_generation() != self.gen:
            raise RuntimeError("param_history entry refers to a device history store that has since been dropped "
                               "(the engine was rebound to a matrix of another size, or hist_clear() was called); "
                               "read param_history before reusing the engine, or construct candidates with a host history "
                               "(n <= 512)")
        vecs = tuple(self.ctx.hist_get([ix], ln)[0] for ix, ln in self.rows)
        return vecs if self.scalar is None else (self.scalar,) + vecs


# _HREF (candstore.HREF): first item of a compact history reference (see _LazyHistory)


# A recorded iterate whose vectors live in the device history store is kept in param_history as an exact tuple of atoms
#     (_HREF, scalar, store generation, index, length [, index, length])
# -- the compact form of a _HistRef.  The cyclic garbage collector stops tracking such a tuple at its first pass, whereas one
# _HistRef instance per candidate and step stays tracked for good: at 6 144 candidates (BASELINE configs[4]) that was half a
# million tracked objects after 20 loop bodies and a full collection of 12-33 ms every third body (tools/gc_probe.py).
def is_history_ref(entry) -> bool:
    return type(entry) is tuple and len(entry) >= 5 and entry[0] is _HREF


class _LazyHistory(list):
    """param_history (AMS:126, 303-304).  Entries recorded by the batched step above n = 512 are references into the
    device-backed history store and become the reference's tuples (lambda, v) / (x,) / (sigma, u, v) when they are
    read; everything else about the list (len, append, iteration, slicing) is a plain list."""
    ctx = None                              # the context whose history store the references point into (set by DeviceEngine.attach)

    def _get(self, i):
        e = list.__getitem__(self, i)
        if isinstance(e, _HistRef):
            e = e.resolve()
            list.__setitem__(self, i, e)
        elif is_history_ref(e):
            rows = tuple((e[k], e[k + 1]) for k in range(3, len(e), 2))
            e = _HistRef(self.ctx, e[1], rows, e[2]).resolve()
            list.__setitem__(self, i, e)
        return e

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._get(k) for k in range(*i.indices(len(self)))]
        return self._get(i if i >= 0 else i + len(self))

    def __iter__(self):
        for k in range(len(self)):
            yield self._get(k)


# ==========================================================================================
# SolutionCandidate (AMS:107-337)
# ==========================================================================================
class SolutionCandidate:
    _candidate_id_counter = 0

    class State(Enum):
        EXPLORING = 1
        REFINING = 2
        STUCK = 3
        CONVERGED = 4
        RETIRED = 5

    def __init__(self, problem_matrix, problem_type, N_diag, initial_lambda=None, initial_v=None, initial_x=None,
                 initial_u=None, initial_sigma=None, initial_weight=0.01, *, engine=None, record_history=None):
        self.id = SolutionCandidate._candidate_id_counter
        SolutionCandidate._candidate_id_counter += 1
        self.N_diag = N_diag
        self.M_rows, self.M_cols = problem_matrix.shape
        self.problem_type = problem_type
        self.problem_matrix = problem_matrix
        # host mirrors of the device rows; _hv = POP_X row (v_k | x_k | right_v_k), _hu = POP_U row (u_k)
        self._hv = None
        self._hu = None
        self._slot = None
        self._engine = None
        # param_history records every iterate like the reference (record_history=False: residual_history only); above
        # n = 512 the vectors stay on the device (history store) until somebody reads them
        self._record_init = True if record_history is None else bool(record_history)      # moves into the store at attach
        self._lazy_hist = max(self.M_rows, self.M_cols) > 512
        self._ph = _LazyHistory()
        self._rh = []
        # the bookkeeping of AMS:113-126 lives in the engine's structure-of-arrays store under this candidate's slot
        # (candstore.py); the attributes below are views of it
        (engine or DeviceEngine.default()).attach(self)
        self.b_vector = None
        self.lambda_k = initial_lambda
        self.sigma_k = initial_sigma
        self.state = SolutionCandidate.State.EXPLORING
        self.w_k = initial_weight
        self.residual_k = float("inf")
        self.prev_residual = float("inf")
        self.alpha_local_step = GLOBAL_DEFAULT_ALPHA_V_INITIAL
        self.stuck_counter = 0
        self.local_psi_retries_needed = 0
        self.num_resets = 0
        self.initialize_random_solution()                     # overwrites any seeds (AMS:127)

    # ---- views of the structure-of-arrays store (candstore.py) -----------------------------------
    # Each getter hands out the object the reference's attribute would hold (same value, same scalar type) and caches it;
    # each setter keeps the assigned object and mirrors its numeric value for the array code of the engine.
    @property
    def residual_k(self):
        st, s = self._st, self._slot
        o = st.res_obj[s]
        if o is _MISSING:
            x = st.res[s]
            o = st.res_obj[s] = x if st.res_kind[s] == _F_NP else float(x)
        return o

    @residual_k.setter
    def residual_k(self, v):
        st, s = self._st, self._slot
        st.res[s], st.res_kind[s] = _classify_real(v)
        st.res_obj[s] = v

    @property
    def prev_residual(self):
        st, s = self._st, self._slot
        o = st.prev_obj[s]
        if o is _MISSING:
            x = st.prev[s]
            o = st.prev_obj[s] = x if st.prev_kind[s] == _F_NP else float(x)
        return o

    @prev_residual.setter
    def prev_residual(self, v):
        st, s = self._st, self._slot
        st.prev[s], st.prev_kind[s] = _classify_real(v)
        st.prev_obj[s] = v

    @property
    def w_k(self):
        st, s = self._st, self._slot
        o = st.w_obj[s]
        if o is _MISSING:
            x = st.w[s]
            o = st.w_obj[s] = x if st.w_kind[s] == _F_NP else float(x)
        return o

    @w_k.setter
    def w_k(self, v):
        st, s = self._st, self._slot
        st.w[s], st.w_kind[s] = _classify_real(v)
        st.w_obj[s] = v

    @property
    def sigma_k(self):
        st, s = self._st, self._slot
        o = st.sig_obj[s]
        if o is _MISSING:
            x = st.sig[s]
            o = st.sig_obj[s] = x if st.sig_kind[s] == _F_NP else float(x)
        return o

    @sigma_k.setter
    def sigma_k(self, v):
        st, s = self._st, self._slot
        st.sig[s], st.sig_kind[s] = _classify_real(v)
        st.sig_obj[s] = v

    @property
    def alpha_local_step(self):
        st, s = self._st, self._slot
        o = st.alpha_obj[s]
        if o is _MISSING:
            x = st.alpha[s]
            o = st.alpha_obj[s] = np.complex128(x) if st.alpha_kind[s] == _C_NP else float(x)
        return o

    @alpha_local_step.setter
    def alpha_local_step(self, v):
        st, s = self._st, self._slot
        t = type(v)
        if t is np.complex128 and v.imag == 0.0:
            st.alpha[s], st.alpha_kind[s] = v.real, _C_NP
        elif t is float:
            st.alpha[s], st.alpha_kind[s] = v, _C_PY
        else:
            st.alpha_kind[s] = _EXACT
            try:
                st.alpha[s] = complex(v).real
            except (TypeError, ValueError):
                st.alpha[s] = np.nan
        st.alpha_obj[s] = v

    @property
    def lambda_k(self):
        st, s = self._st, self._slot
        o = st.lam_obj[s]
        if o is _MISSING:
            z = st.lam[s]
            o = st.lam_obj[s] = z if st.lam_kind[s] == _C_NP else complex(z)
        return o

    @lambda_k.setter
    def lambda_k(self, v):
        st, s = self._st, self._slot
        t = type(v)
        if t is np.complex128:
            st.lam[s], st.lam_kind[s] = v, _C_NP
        elif t is complex:
            st.lam[s], st.lam_kind[s] = v, _C_PY
        else:
            st.lam_kind[s] = _EXACT
            try:
                st.lam[s] = complex(v)
            except (TypeError, ValueError):
                st.lam[s] = np.nan
        st.lam_obj[s] = v

    @property
    def state(self):
        return _STATE_BY_CODE[self._st.state[self._slot]]

    @state.setter
    def state(self, v):
        self._st.state[self._slot] = v.value

    @property
    def stuck_counter(self):
        return int(self._st.stuck[self._slot])

    @stuck_counter.setter
    def stuck_counter(self, v):
        self._st.stuck[self._slot] = v

    @property
    def local_psi_retries_needed(self):
        return int(self._st.retries[self._slot])

    @local_psi_retries_needed.setter
    def local_psi_retries_needed(self, v):
        self._st.retries[self._slot] = v

    @property
    def num_resets(self):
        return int(self._st.resets[self._slot])

    @num_resets.setter
    def num_resets(self, v):
        self._st.resets[self._slot] = v

    @property
    def b_vector(self):
        return self._st.b_obj[self._slot]

    @b_vector.setter
    def b_vector(self, v):
        self._st.b_obj[self._slot] = v

    @property
    def _record(self):
        return bool(self._st.records[self._slot])

    @_record.setter
    def _record(self, v):
        self._st.records[self._slot] = bool(v)

    @property
    def _host_valid(self):                 # host mirrors are current
        return bool(self._st.host_valid[self._slot])

    @_host_valid.setter
    def _host_valid(self, v):
        self._st.host_valid[self._slot] = v

    @property
    def _dev_valid(self):                  # device rows are current
        return bool(self._st.dev_valid[self._slot])

    @_dev_valid.setter
    def _dev_valid(self, v):
        self._st.dev_valid[self._slot] = v

    # ---- histories (AMS:126, 303-304) ---------------------------------------------------------------
    # Above n = 512 the batched step does not touch the candidate objects: the engine logs one record per step
    # (DeviceEngine._log_history) and the lists are brought up to date when somebody reads them.
    def _replay_history(self):
        log = self._st.hist_log
        n = len(log)
        k = self._hist_seen
        if k < n:
            slot = self._slot
            while k < n:
                log[k].replay(slot, self._rh, self._ph if self._record else None)
                k += 1
            self._hist_seen = n

    @property
    def param_history(self):
        self._replay_history()
        return self._ph

    @param_history.setter
    def param_history(self, v):
        self._replay_history()
        self._ph = v

    @property
    def residual_history(self):
        self._replay_history()
        return self._rh

    @residual_history.setter
    def residual_history(self, v):
        self._replay_history()
        self._rh = v

    # ---- host <-> device mirrors --------------------------------------------------------------
    def _len_v(self):
        return self.M_cols if self.problem_type == ProblemType.SVD else self.N_diag

    def _pull(self):
        if not self._host_valid:
            ctx = self._engine.ctx
            self._hv = ctx.pop_get(POP_X, [self._slot], self._len_v())[0]
            if self.problem_type == ProblemType.SVD:
                self._hu = ctx.pop_get(POP_U, [self._slot], self.M_rows)[0]
            self._host_valid = True

    def _push(self, force=False):
        if force or not self._dev_valid:
            if self._engine._deferred is not None:          # inside MAUS_Solver's spawn loop: pushed with the others (engine.end_deferred_push)
                self._engine._deferred.append(self)
                self._dev_valid = False
                return
            ctx = self._engine.ctx
            if self._hv is not None:
                ctx.pop_put(POP_X, [self._slot], self._hv)
            if self._hu is not None and self.problem_type == ProblemType.SVD:
                ctx.pop_put(POP_U, [self._slot], self._hu)
            self._dev_valid = True

    def _invalidate(self):
        """The device rows were updated by a kernel: host mirrors are stale."""
        self._host_valid = False
        self._dev_valid = True

    def _restore_device(self):
        """Undo a speculative device update from the (pre-step) host mirrors."""
        assert self._hv is not None
        self._host_valid = True
        self._push(force=True)

    def _set_vec(self, name, value):
        self._pull()
        setattr(self, name, None if value is None else np.asarray(value, dtype=np.complex128))
        self._host_valid = True
        self._dev_valid = False

    @property
    def v_k(self):
        if self.problem_type != ProblemType.EIGENVALUE:
            return None
        self._pull()
        return self._hv

    @v_k.setter
    def v_k(self, value):
        if self.problem_type == ProblemType.EIGENVALUE:
            self._set_vec("_hv", value)

    @property
    def x_k(self):
        if self.problem_type != ProblemType.SOLVE_LINEAR_SYSTEM:
            return None
        self._pull()
        return self._hv

    @x_k.setter
    def x_k(self, value):
        if self.problem_type == ProblemType.SOLVE_LINEAR_SYSTEM:
            self._set_vec("_hv", value)

    @property
    def right_v_k(self):
        if self.problem_type != ProblemType.SVD:
            return None
        self._pull()
        return self._hv

    @right_v_k.setter
    def right_v_k(self, value):
        if self.problem_type == ProblemType.SVD:
            self._set_vec("_hv", value)

    @property
    def u_k(self):
        if self.problem_type != ProblemType.SVD:
            return None
        self._pull()
        return self._hu

    @u_k.setter
    def u_k(self, value):
        if self.problem_type == ProblemType.SVD:
            self._set_vec("_hu", value)

    # ---- AMS:129-143 -----------------------------------------------------------------------------
    def initialize_random_solution(self):
        rand_vec_init = lambda N: (np.random.rand(N) + 1j * np.random.rand(N)).astype(np.complex128)

        def norm_rand_vec(v):
            if np.linalg.norm(v) > 1e-10:
                return v / np.linalg.norm(v)
            w = rand_vec_init(v.shape[0])
            return w / np.linalg.norm(rand_vec_init(v.shape[0]))

        if self.problem_type == ProblemType.EIGENVALUE:
            self.v_k = norm_rand_vec(rand_vec_init(self.N_diag))
            self.lambda_k = (random.random() * 5 - 2.5 + 1j * (random.random() * 5 - 2.5))
        elif self.problem_type == ProblemType.SOLVE_LINEAR_SYSTEM:
            self.x_k = norm_rand_vec(rand_vec_init(self.N_diag)) * random.uniform(0.1, 10.0)
        elif self.problem_type == ProblemType.SVD:
            self.u_k = norm_rand_vec(rand_vec_init(self.M_rows))
            self.right_v_k = norm_rand_vec(rand_vec_init(self.M_cols))
            self.sigma_k = 1.0
        self._push(force=True)
        self.param_history.append(self.get_current_solution_params())
        self.residual_history.append(self.residual_k)

    def _record_history(self):
        """AMS:303-304 for a candidate whose vectors are recorded from the host mirrors (n <= 512; the engine refreshed them
        with one transfer, DeviceEngine._log_history).  Larger problems never get here: their steps are logged by the engine."""
        if self._record:
            self._ph.append(self.get_current_solution_params())
        self._rh.append(self.residual_k)

    # ---- AMS:145-331 -----------------------------------------------------------------------------
    def update_solution_step(self, current_matrix_A, b_vector=None, strat_params=None, global_knowledge=None):
        if _is_sparse(current_matrix_A) or global_knowledge.get("is_sparse_problem", False):
            raise NotImplementedError("sparse problems are outside the MI355X hot path (dense only)")
        self._engine.step([self], current_matrix_A, b_vector, strat_params, global_knowledge)

    def get_current_solution_params(self):                      # AMS:333-337
        if self.problem_type == ProblemType.EIGENVALUE:
            return (self.lambda_k, self.v_k)
        elif self.problem_type == ProblemType.SOLVE_LINEAR_SYSTEM:
            return (self.x_k,)
        elif self.problem_type == ProblemType.SVD:
            return (self.sigma_k, self.u_k, self.right_v_k)
        return None


_STATE_BY_CODE = (None,) + tuple(SolutionCandidate.State)          # State.value -> member (candstore.py keeps the value)
_CONV, _RETIRED = SolutionCandidate.State.CONVERGED.value, SolutionCandidate.State.RETIRED.value


# ==========================================================================================
# MAUS_Solver (AMS:340-608)
# ==========================================================================================
class MAUS_Solver:
    def __init__(self, problem_matrix, problem_type, b_vector=None, initial_num_candidates=None,
                 global_convergence_tol=1e-8, *, device=0, pert_mode="auto", gmres_compat="rtol",
                 record_history=None, comm=None, quiet=False, engine=None, gram_min=8, cond_exact_max=1024,
                 diag_info=None, eigh_mode="auto"):
        if _is_sparse(problem_matrix):
            raise NotImplementedError("sparse problems are outside the MI355X hot path (dense only)")
        self.M = problem_matrix.astype(np.complex128)                                   # AMS:343
        self.N_rows, self.N_cols = self.M.shape
        self.N_diag = self.N_rows
        self.problem_type = problem_type
        self.b = b_vector.astype(np.complex128) if b_vector is not None else None
        # start-up diagnostics (AMS:374-404): above `cond_exact_max` the condition number comes from the GPU estimator
        # (engine.estimate_condition_number) unless it lands near one of the thresholds it feeds; `engine` (the test
        # seam) keeps the reference's exact computation
        self._cond_device = device if (engine is None and cond_exact_max is not None) else None
        self._cond_exact_max = cond_exact_max
        # `engine` is a test seam (tests/fake_ctx.py drives the host logic without a GPU); product
        # code never passes it, and DeviceEngine() raises if libmaus_hip / the device is missing
        self.engine = engine if engine is not None else DeviceEngine(device=device, pert_mode=pert_mode,
                                                                     gmres_compat=gmres_compat, comm=comm, eigh_mode=eigh_mode)
        # `diag_info`: start-up diagnostics of the same matrix taken from an earlier solver (bench side runs)
        if diag_info is not None:
            self.diag_info = dict(diag_info)
        elif comm is None or comm.world == 1:
            self.diag_info = self._diagnose_matrix_initial(self.M)
        else:
            # Sharded run: rank 0 alone diagnoses the matrix (AMS:374-404: symmetry checks, condition number -- an SVD or,
            # for a Hermitian eigenproblem, the eigh the shortcut needs anyway) with the node's BLAS threads, and broadcasts
            # the result: N ranks must not repeat an O(n^3) host computation, and the strategy -- with it the sequence of
            # collectives -- cannot diverge between ranks.
            # (root_call: an exception on rank 0 -- device memory for the decomposition, MemoryError for an embedding -- is
            # raised on every rank instead of leaving the others inside the broadcasts that follow)
            def diagnose():
                with comm.all_blas_threads():
                    di = self._diagnose_matrix_initial(self.M)
                di["_eigh_seed"] = "_eigh_seed" in self.__dict__
                return di
            self.diag_info = comm.root_call(diagnose)
            if self.diag_info.pop("_eigh_seed", False):
                self.engine.bind_matrix(self.M)
                seed = self.__dict__.pop("_eigh_seed", None)
                self.engine.seed_eigh_distributed(self.M, *(seed[1:] if seed is not None else (None, None)))
        self.is_sparse_problem_init = self.diag_info["is_sparse_init"]
        self.cond_number = self.diag_info["condition_number"]
        self.problem_knowledge = {
            "matrix_type": "Dense", "spectrum_hint": "Unknown", "numerical_stability_state": "Stable",
            "local_solver_preference": "direct_solve", "effective_rank_SVD": min(self.N_rows, self.N_cols),
            "true_matrix_is_singular": self.diag_info["is_singular"],
            "is_sparse_problem": self.is_sparse_problem_init,
            "is_hermitian": self.diag_info.get("is_hermitian", False),
            "is_complex_symmetric": self.diag_info.get("is_complex_symmetric", False),
        }
        if self.problem_knowledge["is_sparse_problem"]:
            raise NotImplementedError("matrices < 25% dense take the reference's sparse path, which is out of scope")
        self.strat_params = {
            "overall_psi_aggression_factor": 1.0, "max_psi_retries": GLOBAL_MAX_PSI_ATTEMPTS,
            "min_survival_weight": GLOBAL_MIN_WEIGHT_TO_SURVIVE_PRUNE, "spawn_rate_multiplier": 1.0,
            "convergence_tolerance": global_convergence_tol, "current_convergence_threshold": global_convergence_tol,
        }
        self._set_initial_strategy()
        self.engine.bind_matrix(self.M)
        seed = self.__dict__.pop("_eigh_seed", None)
        if seed is not None:
            self.engine.seed_eigh(*seed)
        self._record_history = record_history
        # distinctness / redundancy tests (AMS:432-451, 509-520): with at least `gram_min` converged candidates the
        # pairwise np.vdot calls are replaced by one device Gram block and vectorised comparisons (same greedy order,
        # thresholds and arithmetic; SURVEY f-2)
        self.gram_min = gram_min
        self._quiet = quiet
        initial_num_candidates = initial_num_candidates if initial_num_candidates is not None else (self.N_diag * 3)
        if self.problem_type == ProblemType.SVD:
            initial_num_candidates = max(initial_num_candidates, min(self.N_rows, self.N_cols) * 3)
        self.engine.ctx.pop_reserve(initial_num_candidates + 64)
        self.candidates = [self._new_candidate() for _ in range(initial_num_candidates)]
        SolutionCandidate._candidate_id_counter = initial_num_candidates              # AMS:368
        if not quiet:
            print(f"MAUS Initialized with {initial_num_candidates} candidates for {self.problem_type.name} "
                  f"(Dims={self.N_rows}x{self.N_cols}).")
            print(f"Initial matrix diagnostics: Cond={self.cond_number:.2e}, MatrixType={self.problem_knowledge['matrix_type']}, "
                  f"Hermitian={self.problem_knowledge['is_hermitian']}. Stability: {self.problem_knowledge['numerical_stability_state']}.")
        self.landscape_energy = 1.0
        self.avg_residual = 1.0
        self.avg_stuckness = 0.0
        self.num_distinct_converged_solutions = 0
        self.converged_solutions = []
        self.true_solution = None
        self.candidate_steps = 0

    def _new_candidate(self, **kw):
        return SolutionCandidate(self.M, self.problem_type, self.N_diag, engine=self.engine,
                                 record_history=self._record_history, **kw)

    # ---- AMS:374-404 (ndarray branch) ---------------------------------------------------------
    def _diagnose_matrix_initial(self, matrix):
        diag_info = {"is_hermitian": False, "is_complex_symmetric": False, "is_sparse_init": False,
                     "condition_number": np.inf, "is_singular": False}
        if isinstance(matrix, np.ndarray):
            diag_info["is_sparse_init"] = (np.count_nonzero(matrix) / matrix.size) < 0.25 if matrix.size > 0 else False
            try:
                if matrix.ndim == 2 and matrix.shape[0] == matrix.shape[1]:
                    if _allclose_to_transpose(matrix, conj=True):              # np.allclose(M, M.conj().T), AMS:381
                        diag_info["is_hermitian"] = True
                    if _allclose_to_transpose(matrix, conj=False):             # np.allclose(M, M.T), AMS:383
                        diag_info["is_complex_symmetric"] = True
            except Exception:
                pass
        cond_num_val = np.inf
        is_singular_val = False
        if (not diag_info["is_sparse_init"] and isinstance(matrix, np.ndarray) and matrix.ndim == 2
                and matrix.shape[0] == matrix.shape[1] and matrix.size > 0):
            try:
                cond_num_val = None
                # A Hermitian eigenproblem whose decomposition runs on the device (1.1 s at n = 8192, and the shortcut of AMS:161
                # needs it anyway) takes its condition number from the eigenvalues at once: the estimator's six LU
                # factorisations would cost four times as much (r03)
                herm_first = (diag_info["is_hermitian"] and self.problem_type == ProblemType.EIGENVALUE
                              and getattr(self, "_cond_device", None) is not None and matrix.shape[0] > self._cond_exact_max
                              and self.engine.use_device_eigh(matrix.shape[0]) and np.all(np.isfinite(matrix)))
                if not herm_first and getattr(self, "_cond_device", None) is not None and matrix.shape[0] > self._cond_exact_max:
                    from .engine import estimate_condition_number
                    kappa, trusted = estimate_condition_number(matrix, device=self._cond_device)
                    diag_info["condition_number_estimate"] = kappa
                    if trusted:
                        cond_num_val = kappa
                diag_info["condition_number_is_estimate"] = cond_num_val is not None
                if (cond_num_val is None and diag_info["is_hermitian"] and self.problem_type == ProblemType.EIGENVALUE
                        and getattr(self, "_cond_device", None) is not None and matrix.shape[0] > self._cond_exact_max):
                    # (or, host decomposition: the estimate fell into the guard band of a threshold.)  A Hermitian eigenproblem
                    # decomposes the matrix anyway (AMS:161, once per matrix here), and sigma_i = |lambda_i|: take the 2-norm condition number
                    # from the eigenvalues and hand the decomposition to the engine instead of running an SVD on top
                    # (44 s + 70 s at n = 8192, profiles/r02_c4_hermitian_8192_end_to_end.txt)
                    import scipy.linalg as sla
                    try:
                        evals = None
                        if self.engine.use_device_eigh(matrix.shape[0]):
                            from ._cabi import MausHipError
                            try:
                                evals, evecs = self.engine.device_eigh(matrix), None   # V stays on the device
                            except MausHipError as err:                                # no device memory for the work copies, ...
                                print(f"(device eigendecomposition failed: {err}; scipy.linalg.eigh on the host instead)")
                        if evals is None:
                            evals, evecs = sla.eigh(matrix)
                        amax, amin = float(np.abs(evals).max()), float(np.abs(evals).min())
                        with np.errstate(divide="ignore"):
                            cond_num_val = np.float64(amax) / np.float64(amin)
                        self._eigh_seed = (matrix, evals, evecs)
                        diag_info["condition_number_from_eigh"] = True
                    except np.linalg.LinAlgError:
                        cond_num_val = None
                if (cond_num_val is None and getattr(self, "_cond_device", None) is not None and matrix.shape[0] > self._cond_exact_max
                        and self.engine.use_device_eigh(2 * matrix.shape[0]) and np.all(np.isfinite(matrix))):
                    # the estimate fell into the guard band of a threshold: np.linalg.cond's own definition, sigma_max / sigma_min,
                    # with the singular values from the device (a host SVD of a 4096 x 4096 matrix is ~20 s of the start-up)
                    from .engine import singular_values_device
                    sv = singular_values_device(matrix, self._cond_device)
                    with np.errstate(divide="ignore", invalid="ignore"):
                        cond_num_val = np.float64(sv[0]) / np.float64(sv[-1])
                    if np.isnan(cond_num_val):
                        cond_num_val = np.float64(np.inf)
                    diag_info["condition_number_from_device_svd"] = True
                if cond_num_val is None:
                    cond_num_val = np.linalg.cond(matrix)
                if np.isinf(cond_num_val) or cond_num_val > 1e15:
                    is_singular_val = True
            except np.linalg.LinAlgError:
                cond_num_val = np.inf
                is_singular_val = True
        diag_info["condition_number"] = cond_num_val
        diag_info["is_singular"] = is_singular_val
        return diag_info

    # ---- AMS:406-422 ------------------------------------------------------------------------------
    def _set_initial_strategy(self):
        sp_, pk = self.strat_params, self.problem_knowledge
        if self.cond_number > 1e12:
            pk["numerical_stability_state"] = "Critical"
            sp_["overall_psi_aggression_factor"] = 50.0
            sp_["max_psi_retries"] = GLOBAL_MAX_PSI_ATTEMPTS * 2
            sp_["current_convergence_threshold"] = 1e-2
            pk["local_solver_preference"] = "iterative_gmres"
        elif self.cond_number > 1e6:
            pk["numerical_stability_state"] = "Fragile"
            sp_["overall_psi_aggression_factor"] = 10.0
            pk["local_solver_preference"] = "iterative_gmres"
            sp_["current_convergence_threshold"] = 1e-4
        else:
            pk["numerical_stability_state"] = "Stable"
            pk["local_solver_preference"] = "direct_solve"
            sp_["current_convergence_threshold"] = sp_["convergence_tolerance"]
        if self.problem_type == ProblemType.SOLVE_LINEAR_SYSTEM and self.diag_info.get("is_singular", False):
            pk["true_matrix_is_singular"] = True
            pk["local_solver_preference"] = "iterative_gmres"
            sp_["overall_psi_aggression_factor"] = max(sp_["overall_psi_aggression_factor"], 20.0)
        if self.problem_type == ProblemType.SVD:
            if pk["numerical_stability_state"] == "Stable":
                sp_["overall_psi_aggression_factor"] = max(sp_["overall_psi_aggression_factor"], 2.0)
            sp_["current_convergence_threshold"] = max(1e-5, sp_["convergence_tolerance"])

    def _pop_view(self):
        """(store, slots, state codes) of self.candidates: the population as arrays (candstore.py).  The slot array is kept
        between calls and rebuilt when the list is another object or another length than last time (spot-checked at three
        positions: the solver itself only ever replaces the list or appends to it)."""
        cl = self.candidates
        n = len(cl)
        if not n:
            return None, np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.uint8)
        st = cl[0]._st
        cache = self.__dict__.get("_view_cache")
        if (cache is not None and cache[0] is cl and len(cache[1]) == n and cl[0]._slot == cache[1][0]
                and cl[-1]._slot == cache[1][-1] and cl[n // 2]._slot == cache[1][n // 2]):
            slots = cache[1]
        else:
            slots = DeviceEngine._slots(cl)
            self._view_cache = (cl, slots)
        return st, slots, st.state[slots]

    def _prefetch_converged(self, view=None):
        """Host mirrors of the CONVERGED candidates' vectors with one transfer per array (AMS:432 / 510 read every one of them
        through get_current_solution_params; one device-to-host copy per candidate was 40 us each -- 8 ms in the loop body in
        which a few hundred candidates converge together)."""
        st, slots, code = view if view is not None else self._pop_view()
        if st is None:
            return
        ix = np.nonzero((code == _CONV) & ~st.host_valid[slots])[0]
        if ix.size > 1:
            stale = [self.candidates[k] for k in ix.tolist()]
            if all(c._engine is self.engine for c in stale):
                self.engine._bulk_pull(stale, slots[ix])
            else:
                self.engine._bulk_pull([c for c in stale if c._engine is self.engine])

    # ---- AMS:424-475 ------------------------------------------------------------------------------
    def _converged_gram(self, view=None):
        """(position map, |Gram| blocks) over the CONVERGED candidates in list order, or (None, None) when the set is
        small / the problem type has no vector test.  EIG: {'v'}; SVD: {'u', 'v'}."""
        if self.problem_type not in (ProblemType.EIGENVALUE, ProblemType.SVD):
            return None, None
        st, slots, code = view if view is not None else self._pop_view()
        if st is None:
            return None, None
        ix = np.nonzero(code == _CONV)[0]
        if ix.size < max(2, self.gram_min):
            return None, None
        conv = [self.candidates[k] for k in ix.tolist()]
        stale = np.nonzero(~st.dev_valid[slots[ix]])[0].tolist()
        for k in stale:
            conv[k]._push()                              # host-side edits (if any) reach the device rows first
        # The block of the previous call serves this one when its candidates are a subset (AMS:504-527 retires, the diagnostics
        # of the next iteration look at the survivors: same vectors -- a CONVERGED candidate is not stepped again -- and the same
        # products, entry for entry).  Keyed by candidate id; a host-side edit of a vector (a push above) drops it.
        cache = self.__dict__.get("_gram_cache")
        if cache is not None and not stale:
            pos = cache[0]
            sel = [pos.get(c.id, -1) for c in conv]
            if min(sel) >= 0:
                sel = np.asarray(sel, dtype=np.int64)
                blocks = {k: b[np.ix_(sel, sel)] for k, b in cache[1].items()}
                return {id(c): i for i, c in enumerate(conv)}, blocks
        eng = self.engine
        if self.problem_type == ProblemType.EIGENVALUE:
            blocks = {"v": np.abs(eng.d_gram(conv, POP_X, self.N_diag))}
        else:
            blocks = {"u": np.abs(eng.d_gram(conv, POP_U, self.N_rows)), "v": np.abs(eng.d_gram(conv, POP_X, self.N_cols))}
        self._gram_cache = ({c.id: i for i, c in enumerate(conv)}, blocks)
        return {id(c): i for i, c in enumerate(conv)}, blocks

    def _update_global_diagnostics(self, iteration):
        C = SolutionCandidate.State
        view = self._pop_view()
        st, slots, code = view
        self._prefetch_converged(view)
        gpos, gram = self._converged_gram(view)
        acc_pos, acc_key = [], []                          # Gram positions / lambda (sigma) of the accepted solutions
        total_active_candidates = len(self.candidates)
        sum_residuals = 0.0
        sum_stuck_counters = 0
        num_converged_all_types = 0
        self.num_distinct_converged_solutions = 0
        self.converged_solutions = []
        current_sigma_magnitudes = []
        thr = self.strat_params["current_convergence_threshold"]
        max_s = None
        # the reference walks the whole population (AMS:429-462); only its CONVERGED members do more than add to two sums
        for k in np.nonzero(code == _CONV)[0].tolist():
            c = self.candidates[k]
            num_converged_all_types += 1
            current_tuple = c.get_current_solution_params()
            is_distinct = True
            if current_tuple is None or any(p is None for p in current_tuple):
                continue
            if self.problem_type == ProblemType.EIGENVALUE and gram is not None:
                if acc_pos:
                    s_lam = np.asarray(acc_key)
                    close = np.abs(current_tuple[0] - s_lam) < (GLOBAL_LAMBDA_SIMILARITY_TOL + np.abs(s_lam) * 1e-6)
                    if close.any() and (close & (gram["v"][gpos[id(c)], acc_pos] > GLOBAL_VECTOR_SIMILARITY_TOL)).any():
                        is_distinct = False
            elif self.problem_type == ProblemType.EIGENVALUE:
                for s_item in self.converged_solutions:
                    s_lam, s_vec = s_item[0], s_item[1]
                    effective_tol = GLOBAL_LAMBDA_SIMILARITY_TOL + np.abs(s_lam) * 1e-6
                    if (np.abs(current_tuple[0] - s_lam) < effective_tol
                            and np.abs(np.vdot(current_tuple[1], s_vec)) > GLOBAL_VECTOR_SIMILARITY_TOL):
                        is_distinct = False
                        break
            elif self.problem_type == ProblemType.SOLVE_LINEAR_SYSTEM:
                if (len(self.converged_solutions) > 0 and
                        np.linalg.norm(current_tuple[0] - self.converged_solutions[0][0])
                        < self.strat_params["convergence_tolerance"] * 100):
                    is_distinct = False
            elif self.problem_type == ProblemType.SVD:
                if max_s is None:                  # AMS:444: the same value for every converged candidate of this pass
                    sg = st.sig[slots]             # (sigma_k of every candidate, NaN where it is None)
                    with np.errstate(invalid="ignore"):
                        pos_sig = sg[sg > 0]
                    max_s = pos_sig.max() if pos_sig.size else 1.0
                if current_tuple[0].real / max_s < GLOBAL_SIGMA_SIMILARITY_TOL_REL:
                    is_distinct = False
                if is_distinct and gram is not None:
                    if acc_pos:
                        s_sig = np.asarray(acc_key)
                        close = np.abs(current_tuple[0] - s_sig) < np.maximum(GLOBAL_SIGMA_SIMILARITY_TOL_ABS, s_sig * GLOBAL_SIGMA_SIMILARITY_TOL_REL)
                        i = gpos[id(c)]
                        if close.any() and (close & (gram["u"][i, acc_pos] > GLOBAL_VECTOR_SIMILARITY_TOL)
                                            & (gram["v"][i, acc_pos] > GLOBAL_VECTOR_SIMILARITY_TOL)).any():
                            is_distinct = False
                elif is_distinct:
                    for s_item in self.converged_solutions:
                        s_sigma, s_u, s_v = s_item
                        if (np.abs(current_tuple[0] - s_sigma) < max(GLOBAL_SIGMA_SIMILARITY_TOL_ABS, s_sigma * GLOBAL_SIGMA_SIMILARITY_TOL_REL)
                                and np.abs(np.vdot(current_tuple[1], s_u)) > GLOBAL_VECTOR_SIMILARITY_TOL
                                and np.abs(np.vdot(current_tuple[2], s_v)) > GLOBAL_VECTOR_SIMILARITY_TOL):
                            is_distinct = False
                            break
                current_sigma_magnitudes.append(current_tuple[0].real)
            if is_distinct:
                self.converged_solutions.append(current_tuple)
                self.num_distinct_converged_solutions += 1
                if gram is not None:
                    acc_pos.append(gpos[id(c)])
                    acc_key.append(current_tuple[0])
        # AMS:463-465 over the candidates that are neither CONVERGED nor RETIRED, summed in list order like the reference's
        # loop (cumsum adds left to right; np.sum would add pairwise and round differently)
        if st is not None:
            act = (code != _CONV) & (code != _RETIRED)
            if act.any():
                asl = slots[act]
                r = st.res[asl]
                fin = np.isfinite(r)
                total = np.cumsum(np.where(fin, r, thr * 100))[-1]
                # (the sum is np.float64 once a residual -- np.float64 after a candidate's first step -- has been added)
                sum_residuals = total if bool((fin & (st.res_kind[asl] != _F_PY)).any()) else float(total)
                sum_stuck_counters = int(st.stuck[asl].sum())
        non_conv_retired_count = max(1, total_active_candidates - num_converged_all_types)
        self.avg_residual = sum_residuals / non_conv_retired_count
        self.avg_stuckness = sum_stuck_counters / non_conv_retired_count
        norm_avg_res = self.avg_residual / (thr * 10)
        norm_avg_stuck = self.avg_stuckness / (GLOBAL_MAX_STUCK_FOR_RETIREMENT * 2)
        target_sols_N_global = self.N_diag
        if self.problem_type == ProblemType.SOLVE_LINEAR_SYSTEM:
            target_sols_N_global = 1
        elif self.problem_type == ProblemType.SVD:
            if len(current_sigma_magnitudes) > 1:
                sorted_sigmas = sorted([s for s in current_sigma_magnitudes if s > GLOBAL_SIGMA_SIMILARITY_TOL_ABS], reverse=True)
                if sorted_sigmas:
                    max_sigma_val = sorted_sigmas[0]
                    rank_detected = sum(1 for s_val in sorted_sigmas if s_val / max_sigma_val > GLOBAL_SIGMA_SIMILARITY_TOL_REL)
                    self.problem_knowledge["effective_rank_SVD"] = min(
                        rank_detected if rank_detected > 0 else 1, min(self.N_rows, self.N_cols),
                        max(1, self.problem_knowledge.get("effective_rank_SVD", 1)))
            target_sols_N_global = self.problem_knowledge.get("effective_rank_SVD", min(self.N_rows, self.N_cols))
        norm_missing_sols = (target_sols_N_global - self.num_distinct_converged_solutions) / max(1, target_sols_N_global)
        self.landscape_energy = max(0.0, min(1.0, (norm_avg_res * 0.4) + (norm_avg_stuck * 0.3) + (norm_missing_sols * 0.3)))
        if self.avg_stuckness > GLOBAL_MAX_STUCK_FOR_RETIREMENT * 0.5:
            self.problem_knowledge["numerical_stability_state"] = "Critical"
        elif self.avg_stuckness > GLOBAL_MAX_STUCK_FOR_PRUNING * 0.5:
            self.problem_knowledge["numerical_stability_state"] = "Fragile"
        else:
            self.problem_knowledge["numerical_stability_state"] = "Stable"

    # ---- AMS:477-501 ------------------------------------------------------------------------------
    def _adjust_global_strategy(self, iteration):
        sp_, pk = self.strat_params, self.problem_knowledge
        stab = pk["numerical_stability_state"]
        tol = sp_["convergence_tolerance"]
        if self.landscape_energy > 0.6 and stab == "Critical":
            pk["local_solver_preference"] = "iterative_gmres"
            sp_["overall_psi_aggression_factor"] = min(200.0, sp_["overall_psi_aggression_factor"] * 1.1)
            sp_["spawn_rate_multiplier"] = min(10.0, sp_["spawn_rate_multiplier"] * 1.2)
            sp_["current_convergence_threshold"] = max(tol * 50, sp_["current_convergence_threshold"] * 1.05)
        elif self.landscape_energy > 0.4 and stab == "Fragile":
            pk["local_solver_preference"] = "iterative_gmres"
            sp_["overall_psi_aggression_factor"] = min(50.0, sp_["overall_psi_aggression_factor"] * 1.05)
            sp_["spawn_rate_multiplier"] = min(5.0, sp_["spawn_rate_multiplier"] * 1.1)
            sp_["current_convergence_threshold"] = max(tol * 5, sp_["current_convergence_threshold"] * 1.02)
        elif self.landscape_energy < 0.2 and stab == "Stable":
            pk["local_solver_preference"] = "direct_solve"
            sp_["overall_psi_aggression_factor"] = max(1.0, sp_["overall_psi_aggression_factor"] * 0.9)
            sp_["spawn_rate_multiplier"] = max(0.01, sp_["spawn_rate_multiplier"] * 0.9)
            sp_["current_convergence_threshold"] = max(tol, sp_["current_convergence_threshold"] * 0.9)
        sp_["overall_psi_aggression_factor"] = max(1.0, min(200.0, sp_["overall_psi_aggression_factor"]))
        sp_["spawn_rate_multiplier"] = max(0.01, min(10.0, sp_["spawn_rate_multiplier"]))
        sp_["current_convergence_threshold"] = max(tol, min(1.0, sp_["current_convergence_threshold"]))

    # ---- AMS:504-549 ------------------------------------------------------------------------------
    def _manage_candidates(self, iteration):
        C = SolutionCandidate.State
        view = self._pop_view()
        st, slots, code = view
        cl = self.candidates
        survivors = []
        if st is not None:
            # sorted(key=lambda x: (-x.w_k, x.residual_k if isfinite else inf)), AMS:506: the same stable order from a
            # lexicographic sort of the two key arrays (the per-candidate key tuples were 5 ms per loop body at 6144 candidates)
            res_key = st.res[slots]
            res_key = np.where(np.isfinite(res_key), res_key, np.inf)
            order = np.lexsort((res_key, -st.w[slots]))
            tol = self.strat_params["convergence_tolerance"]
            self._prefetch_converged(view)
            gpos, gram = self._converged_gram(view)
            sur_pos, sur_key = [], []                          # Gram positions / lambda (sigma) of the converged survivors
            conv_survivors = []
            scode = code[order]
            sslots = slots[order]
            retire = np.zeros(len(cl), dtype=bool)             # in sorted order
            # AMS:507-527 only does something for CONVERGED candidates: the greedy redundancy test against the converged
            # survivors in front of them
            for pos in np.nonzero(scode == _CONV)[0].tolist():
                c = cl[order[pos]]
                redundant = False
                if gram is not None:
                    tc = c.get_current_solution_params()
                    if sur_pos and not (tc is None or any(p is None for p in tc)):
                        i = gpos[id(c)]
                        key = np.asarray(sur_key)
                        if self.problem_type == ProblemType.EIGENVALUE:
                            close = np.abs(tc[0] - key) < (GLOBAL_LAMBDA_SIMILARITY_TOL + np.abs(key) * 1e-6)
                            redundant = bool(close.any() and (close & (gram["v"][i, sur_pos] > GLOBAL_VECTOR_SIMILARITY_TOL)).any())
                        else:
                            live = ~(key.real < GLOBAL_SIGMA_SIMILARITY_TOL_ABS / 100)
                            close = np.abs(tc[0] - key) < np.maximum(GLOBAL_SIGMA_SIMILARITY_TOL_ABS, key * GLOBAL_SIGMA_SIMILARITY_TOL_REL)
                            redundant = bool((live & close & (gram["u"][i, sur_pos] > GLOBAL_VECTOR_SIMILARITY_TOL)
                                              & (gram["v"][i, sur_pos] > GLOBAL_VECTOR_SIMILARITY_TOL)).any())
                else:
                    for s_c in conv_survivors:
                        tc, ts = c.get_current_solution_params(), s_c.get_current_solution_params()
                        if tc is None or ts is None or any(p is None for p in tc) or any(p is None for p in ts):
                            continue
                        if self.problem_type == ProblemType.EIGENVALUE:
                            if (np.abs(tc[0] - ts[0]) < (GLOBAL_LAMBDA_SIMILARITY_TOL + np.abs(ts[0]) * 1e-6)
                                    and np.abs(np.vdot(tc[1], ts[1])) > GLOBAL_VECTOR_SIMILARITY_TOL):
                                redundant = True
                                break
                        elif self.problem_type == ProblemType.SOLVE_LINEAR_SYSTEM:
                            if np.linalg.norm(tc[0] - ts[0]) < tol * 10:
                                redundant = True
                                break
                        elif self.problem_type == ProblemType.SVD:
                            if ts[0].real < GLOBAL_SIGMA_SIMILARITY_TOL_ABS / 100:
                                redundant = False
                            elif (np.abs(tc[0] - ts[0]) < max(GLOBAL_SIGMA_SIMILARITY_TOL_ABS, ts[0] * GLOBAL_SIGMA_SIMILARITY_TOL_REL)
                                  and np.abs(np.vdot(tc[1], ts[1])) > GLOBAL_VECTOR_SIMILARITY_TOL
                                  and np.abs(np.vdot(tc[2], ts[2])) > GLOBAL_VECTOR_SIMILARITY_TOL):
                                redundant = True
                                break
                if redundant:
                    retire[pos] = True
                else:
                    conv_survivors.append(c)
                    if gram is not None:
                        ts = c.get_current_solution_params()
                        if not (ts is None or any(p is None for p in ts)):
                            sur_pos.append(gpos[id(c)])
                            sur_key.append(ts[0])
            # AMS:528-531 for everybody else: weight below the survival threshold or stuck for too long
            other = (scode != _CONV) & (scode != _RETIRED)
            retire |= other & ((st.w[sslots] < self.strat_params["min_survival_weight"])
                               | (st.stuck[sslots] >= GLOBAL_MAX_STUCK_FOR_RETIREMENT))
            st.state[sslots[retire]] = _RETIRED
            keep = ~retire & (scode != _RETIRED)
            survivors = [cl[k] for k in order[keep].tolist()]
            self._view_cache = (survivors, sslots[keep])
        self.candidates = survivors
        target = self.N_diag
        if self.problem_type == ProblemType.SOLVE_LINEAR_SYSTEM:
            target = 1
        elif self.problem_type == ProblemType.SVD:
            target = self.problem_knowledge.get("effective_rank_SVD", min(self.N_rows, self.N_cols))
        desired_pop_base = max(5, int(self.N_diag * 1.5 if self.problem_type == ProblemType.SOLVE_LINEAR_SYSTEM else self.N_diag * 2))
        if self.problem_type == ProblemType.SVD:
            desired_pop_base = max(desired_pop_base, int(target * 2.5))
        num_to_spawn = max(0, desired_pop_base - len(self.candidates)) + max(0, target - self.num_distinct_converged_solutions)
        num_to_spawn = min(int(num_to_spawn * self.strat_params["spawn_rate_multiplier"]), self.N_diag * 2, 15)
        self.engine.begin_deferred_push()                   # the spawns' vectors reach the device in one transfer
        try:
            for _ in range(max(0, num_to_spawn)):
                kw = {}
                if self.num_distinct_converged_solutions > 0 and self.landscape_energy < 0.8 and self.converged_solutions:
                    base_sol_tuple = random.choice(self.converged_solutions)                       # E7
                    if base_sol_tuple is None or any(p is None for p in base_sol_tuple):
                        continue
                    if self.problem_type == ProblemType.EIGENVALUE:
                        kw["initial_lambda"] = base_sol_tuple[0] + (random.random() * 0.1 - 0.05 + 1j * (random.random() * 0.1 - 0.05)) * (0.1 + self.landscape_energy)
                        v_pert = (np.random.rand(self.N_diag) - 0.5 + 1j * (np.random.rand(self.N_diag) - 0.5)) * (0.1 + self.landscape_energy)
                        new_v = base_sol_tuple[1] + v_pert
                        norm_new_v = np.linalg.norm(new_v)
                        kw["initial_v"] = new_v / norm_new_v if norm_new_v > 1e-9 else \
                            (np.random.rand(self.N_diag) + 1j * np.random.rand(self.N_diag)) / np.sqrt(self.N_diag)
                new_candidate = self._new_candidate(**kw, initial_weight=0.01)
                new_candidate.alpha_local_step = GLOBAL_DEFAULT_ALPHA_V_INITIAL * (1 + self.strat_params["overall_psi_aggression_factor"] / 10.0)
                self.candidates.append(new_candidate)
        finally:
            self.engine.end_deferred_push()
        cache = self.__dict__.get("_view_cache")
        if cache is not None and cache[0] is self.candidates and len(cache[1]) < len(self.candidates):
            born = np.asarray([c._slot for c in self.candidates[len(cache[1]):]], dtype=np.int64)
            self._view_cache = (self.candidates, np.concatenate([cache[1], born]))

    # ---- the loop body AMS:573-577 ------------------------------------------------------------------
    def step_population(self):
        """The hot loop `for candidate in self.candidates: update_solution_step(...)`, batched."""
        C = SolutionCandidate.State
        st, slots, code = self._pop_view()
        ix = np.nonzero((code != _CONV) & (code != _RETIRED))[0]
        active = [self.candidates[k] for k in ix.tolist()]
        self.engine.step(active, self.M, self.b, self.strat_params, self.problem_knowledge, slots[ix])
        self.candidate_steps += len(active)
        return len(active)

    def loop_body(self, iteration):
        self._update_global_diagnostics(iteration)
        self._adjust_global_strategy(iteration)
        n = self.step_population()
        self._manage_candidates(iteration)
        return n

    # ---- AMS:551-608 ------------------------------------------------------------------------------
    @property
    def true_solution(self):
        """AMS:553-570: the reference answer of evolve()'s closing comparison (None before evolve() / if it failed)."""
        if self._true_solution is _DEFERRED:
            import time
            t0 = time.perf_counter()
            self._true_solution = self._reference_solution()
            if time.perf_counter() - t0 > 5.0:
                print(f"(reference solution for the closing comparison: {time.perf_counter() - t0:.1f} s on the host; "
                      f"evolve(reference_check=False) skips it)")
        return self._true_solution

    @true_solution.setter
    def true_solution(self, value):
        self._true_solution = value

    def _reference_solution(self):
        """The reference's reporting prologue (AMS:554-570): a reference answer to compare the final report with.  Linear
        systems above n = 512 on the device LU, Hermitian spectra and singular values from n = 1536 up through the device
        tridiagonalisation (SURVEY f-4); general eigenvalues by SciPy on the host (O(n^3))."""
        import scipy.linalg as sla
        try:
            if self.M.size == 0:
                raise ValueError("Matrix is empty.")
            if self.problem_type == ProblemType.EIGENVALUE:
                if self.N_rows != self.N_cols:
                    raise ValueError("Non-square matrix for Eigenvalue.")
                if self.problem_knowledge.get("is_hermitian", False) and self.engine.use_device_eigh(self.N_rows):
                    # Hermitian: the spectrum of the tridiagonal matrix the device reduces M to (csrc/herm.hip) instead of a
                    # general QR iteration on the host -- eigenvalues only (bisection on the device), real, in eigvals()'s sorted order
                    from .engine import tridiagonal_eigenvalues
                    from ._cabi import MausHipError
                    try:
                        self.engine.bind_matrix(self.M)
                        d, e = self.engine.ctx.herm_tridiag()
                        try:
                            return np.sort(tridiagonal_eigenvalues(self.engine.ctx, d, e)).astype(np.complex128)
                        finally:
                            if hasattr(self.engine.ctx, "herm_release"):
                                self.engine.ctx.herm_release()            # eigenvalues only: no back-transformation follows
                    except MausHipError as err:                           # e.g. no device memory for the n x n work copies
                        print(f"(device reduction for the reference eigenvalues failed: {err}; SciPy on the host instead)")
                vals = sla.eigvals(self.M)
                vals.sort()
                return vals
            if self.problem_type == ProblemType.SOLVE_LINEAR_SYSTEM:
                if self.b is None:
                    raise ValueError("b_vector is None.")
                if self.N_rows != self.b.shape[0]:
                    raise ValueError("A,b shape mismatch.")
                if self.N_rows > 512 and hasattr(self.engine.ctx, "lu_solve"):
                    # the batched LU (LAPACK's pivot order) on the device instead of a host zgesv: the same answer to
                    # conditioning, in milliseconds at n = 4096
                    if not (np.all(np.isfinite(self.M)) and np.all(np.isfinite(self.b))):
                        raise ValueError("array must not contain infs or NaNs")
                    x, st = self.engine.ctx.lu_solve(self.M, self.b)
                    if st[0] > 0:
                        raise np.linalg.LinAlgError("Matrix is singular.")
                    self.engine.bind_matrix(self.M)
                    return x[0]
                return sla.solve(self.M, self.b, assume_a="general")
            if self.engine.use_device_eigh(self.N_rows + self.N_cols) and np.all(np.isfinite(self.M)):
                # singular values as the positive eigenvalues of the Hermitian embedding [[0, M], [M^H, 0]] (+-sigma_i and
                # |rows - cols| zeros), reduced to tridiagonal form on the device: the same absolute accuracy eps ||M|| as
                # LAPACK's bidiagonal SVD, without its O(n^3) on the host
                from .engine import singular_values_device
                return singular_values_device(self.M, self.engine.ctx.device).tolist()
            return sorted(sla.svd(self.M, compute_uv=False).tolist(), reverse=True)
        except (np.linalg.LinAlgError, ValueError) as e:
            print(f"NumPy reference calculation failed: {e}.")
            return None

    def _report_residual(self, t):
        """Residual of one reported solution tuple, recomputed from the problem matrix (AMS:594-596)."""
        M = self.M
        if self.problem_type == ProblemType.EIGENVALUE:
            return np.linalg.norm(M @ t[1] - t[0] * t[1])
        if self.problem_type == ProblemType.SOLVE_LINEAR_SYSTEM:
            return np.linalg.norm(M @ t[0] - self.b)
        return np.linalg.norm(M @ t[2] - t[0] * t[1]) + np.linalg.norm(M.conj().T @ t[1] - t[0] * t[2])

    def evolve(self, max_iterations=100, *, reference_check=None):
        """AMS:551-608.  `reference_check` (default on, as in the reference): the "true solution" prologue (AMS:554-570)
        and the closing comparison.  Linear systems above n = 512 are solved by the device LU, Hermitian spectra and singular
        values come from the device tridiagonalisation; general eigenvalues above n = 512 -- SciPy on the host, about a minute
        at n = 4096 -- are computed lazily, when `true_solution` is first read; False skips the whole prologue."""
        print(f"--- Starting MAUS Evolution for {max_iterations} iterations ({self.problem_type.name}) ---")
        self.true_solution = None
        if reference_check is None:
            reference_check = True
        if reference_check:
            if (self.problem_type == ProblemType.EIGENVALUE and self.N_rows == self.N_cols and self.N_rows > REFERENCE_LAZY_MIN
                    and not (self.problem_knowledge.get("is_hermitian", False) and self.engine.use_device_eigh(self.N_rows))):
                # General eigenvalues are the one reference answer that still costs O(n^3) on the host (about a minute at
                # n = 4096): computed when `true_solution` is first read -- by the closing comparison below, which only
                # happens if something converged, or by the caller -- instead of in front of the first iteration.
                self._true_solution = _DEFERRED
            else:
                self.true_solution = self._reference_solution()
        for i in range(max_iterations):
            self.loop_body(i + 1)
            target_sols_disp = self.N_diag
            if self.problem_type == ProblemType.SOLVE_LINEAR_SYSTEM:
                target_sols_disp = 1
            elif self.problem_type == ProblemType.SVD:
                target_sols_disp = self.problem_knowledge.get("effective_rank_SVD", min(self.N_rows, self.N_cols))
            target_sols_final = target_sols_disp       # undefined in the reference (NameError at AMS:583)
            if (i + 1) % 20 == 0 or i == max_iterations - 1:
                print(f"Iter {i+1}/{max_iterations}: Energy={self.landscape_energy:.2f}, AvgRes={self.avg_residual:.2e}, "
                      f"Conv={self.num_distinct_converged_solutions}/{target_sols_disp}, "
                      f"Stab={self.problem_knowledge['numerical_stability_state']}")
            if (self.num_distinct_converged_solutions >= target_sols_final and self.landscape_energy < 0.05
                    and self.avg_residual < self.strat_params["convergence_tolerance"]):
                print(f"MAUS converged early at iteration {i+1}.")
                break
            if i == max_iterations - 1 and self.num_distinct_converged_solutions < target_sols_final:
                print(f"WARNING: Max iterations. Found {self.num_distinct_converged_solutions}/{target_sols_final}.")
        print("--- MAUS Evolution COMPLETE ---")
        print("Final Report:")
        sols = list(self.converged_solutions)
        if self.problem_type == ProblemType.EIGENVALUE:
            sols.sort(key=lambda x: (x[0].real, x[0].imag) if x[0] is not None else (float("inf"), float("inf")))
        elif self.problem_type == ProblemType.SVD:
            sols.sort(key=lambda x: -x[0].real if x[0] is not None else float("-inf"))
        for k, t in enumerate(sols):
            if t is None or any(p is None for p in t):
                print(f"  Solution {k+1}: Invalid")
                continue
            res = self._report_residual(t)
            if self.problem_type == ProblemType.EIGENVALUE:
                print(f"  Eig {k+1}: λ={t[0]:.6e}, Res={res:.2e}")
            elif self.problem_type == ProblemType.SOLVE_LINEAR_SYSTEM:
                print(f"  LinSolve {k+1}: X_norm1={np.linalg.norm(t[0], 1):.6e}, Res={res:.2e}")
            else:
                print(f"  SVD {k+1}: σ={t[0]:.6e}, Res={res:.2e}")
        if self.num_distinct_converged_solutions > 0 and sols and self.true_solution is not None:
            print("--- Comparison to NumPy ---")
            if self.problem_type == ProblemType.EIGENVALUE:
                found = np.array(sorted([t[0] for t in sols if t[0] is not None], key=lambda z: (z.real, z.imag)))
                ref = self.true_solution[:len(found)]
                if found.size > 0 and ref.size > 0:
                    print(f"Mean abs error (eigs): {np.sum(np.abs(found - ref)) / len(found):.2e}")
            elif self.problem_type == ProblemType.SOLVE_LINEAR_SYSTEM:
                if sols[0][0] is not None:
                    err, nref = np.linalg.norm(sols[0][0] - self.true_solution), np.linalg.norm(self.true_solution)
                    print(f"Rel error (X): {err / nref if nref > 1e-10 else err:.2e}")
            else:
                found = np.array(sorted([t[0].real for t in sols if t[0] is not None], reverse=True))
                ref = np.array(self.true_solution[:len(found)])
                if found.size > 0 and ref.size > 0:
                    nref = np.linalg.norm(ref)
                    err = np.linalg.norm(found - ref)
                    print(f"Rel error (sigmas): {err / nref if nref > 1e-10 else err:.2e}")
