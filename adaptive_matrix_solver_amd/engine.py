"""Batched device execution of SolutionCandidate.update_solution_step (AMS:145-331).

The reference steps candidates one by one; each step is independent of the others except
for the ORDER in which the two global RNG streams are consumed.  The
engine therefore runs the numerical phases for a whole run of candidates as batched HIP
calls (libmaus_hip via ctypes) and replays the RNG events in list order on the host:

  E2  collapsed eigenvector re-init before the solve          AMS:259-263
  E3  2 x rand(N,N) per dense solve attempt                   AMS:49
  E4  tiny-norm re-init after the relaxed update              AMS:283
  E5  random re-initialisation after a failed step            AMS:293

A run is executed speculatively assuming the common case (no E2/E4/E5, first attempt
succeeds); the first candidate that deviates is then handled on its own with the exact
sequential semantics (the retry/fallback ladder of AMS:43-104) and the run restarts behind
it.  With the exact perturbation (`pert_mode='uniform'`) later candidates of the run are
recomputed because their draws moved; with `pert_mode='none'` (the 0.15*psi perturbation
dropped, stream still advanced) their results do not depend on the stream and are kept.

No CPU fallback: every matrix-sized operation goes through the C ABI.
"""
from __future__ import annotations

import os
import random as _pyrandom
import weakref

import numpy as np

from . import _cabi
from .candstore import C_NP, C_PY, EXACT, F_NP, F_PY, MISSING, CandidateStore, HistoryRecord
from ._cabi import KIND_EIG, KIND_LINEAR, KIND_SVD, PERT_MT19937, PERT_NONE, PERT_UNIFORM, POP_U, POP_W, POP_X

# reference constants (AMS:16-26) used by the step
PSI_EPSILON_BASE = np.complex128(1e-20)
MAX_PSI_ATTEMPTS = 25
MAX_STUCK_FOR_RETIREMENT = 8
SIGMA_SIMILARITY_TOL_ABS = 1e-6
CONVERGENCE_RESIDUAL_TOL = 1e-8

DIRECT, GMRES = "direct_solve", "iterative_gmres"


import operator as _operator
_SLOT_OF = _operator.attrgetter("_slot")


class _SolveFailed(RuntimeError):
    pass


def _advance_numpy_stream(nwords: int) -> None:
    """Advance the global legacy NumPy stream by `nwords` MT19937 outputs without drawing them."""
    if nwords <= 0:
        return
    st = np.random.get_state()
    key, pos = _cabi.mt19937_jump(st[1], st[2], nwords)
    np.random.set_state((st[0], key, pos, st[3], st[4]))


class _AsyncStreamAdvance:
    """Compute the NumPy-stream advance of a whole run on a worker thread while the GPU executes the
    batch (the jump polynomial x^J mod phi costs ~20 ms for a new J; ctypes releases the GIL).  The
    result is applied only if the run completes without RNG events; otherwise it is discarded and
    the exact sequential replay takes over."""

    def __init__(self, nwords: int):
        import threading
        self.nwords = nwords
        self.state = np.random.get_state()
        self.result = None
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _run(self):
        self.result = _cabi.mt19937_jump(self.state[1], self.state[2], self.nwords)

    def apply(self):
        self.thread.join()
        key, pos = self.result
        st = self.state
        np.random.set_state((st[0], key, pos, st[3], st[4]))

    def discard(self):
        self.thread.join()


def cpu_allowance() -> int:
    """CPUs this process may actually use: the affinity mask, cut down by the cgroup's CPU quota (cpu.max) when there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:                                                         # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:                                                     # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, quota // period))
        except Exception:
            pass
    return max(1, n)


_blas_cap_done = False


def cap_blas_threads() -> None:
    """Keep the host BLAS pools within the CPUs the process is allowed (once per process; MAUS_BLAS_THREADS=0 leaves them
    alone, =k sets k).  OpenBLAS starts one thread per visible core -- 256 on an MI355X host -- and its workers spin for a
    while after every parallel call; under a 16-CPU container quota the orchestration thread then loses the CPU for 50-90 ms
    at a time, at random inside a loop body or inside a library call that is waiting for the GPU (BASELINE configs[2]: one
    17 ms loop body in ten took 90 ms; tools/body_probe.py).  The loop bodies only use BLAS for dot products of length n."""
    global _blas_cap_done
    if _blas_cap_done:
        return
    _blas_cap_done = True
    env = os.environ.get("MAUS_BLAS_THREADS", "auto")
    if env == "0":
        return
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        limit = int(env) if env not in ("auto", "") else cpu_allowance()
        if any(m.get("num_threads", 1) > limit for m in threadpool_info()):
            threadpool_limits(limits=limit)          # not used as a context manager: stays in force
    except Exception:
        pass


TRIDIAG_MIN_GAP = 1e-7                     # device tridiagonal eigenvectors are kept when min gap / ||T|| is at least this ...
TRIDIAG_MAX_RESID = 1e-13                  # ... and max |(T z - lambda z)_k| / ||T|| at most this (device_eigh)
EIGH_DEVICE_MIN = 1536                     # eigh_mode='auto': device reduction / back-transformation from this order up
COND_THRESHOLDS = (1e6, 1e12, 1e15)        # AMS:401, 407-416: the only places the condition number is used
COND_GUARD = 30.0                          # an estimate this close (either side) to a threshold is not trusted


def estimate_condition_number(A, device=0, max_iter=25, rtol=1e-2):
    """2-norm condition estimate of a dense square matrix on the GPU (SURVEY f-3), replacing the full SVD of
    np.linalg.cond (20-30 s at n=4096, minutes at 8192) in MAUS_Solver's start-up diagnostics (AMS:400).

    sigma_max: alternating power steps (the SVD step kernel, AMS:233-242); sigma_min: inverse iteration
    v <- A^-H (A^-1 v) with the batched LU on A and on A^H (two scratch contexts), 1/||A^-1 v|| -> sigma_min.
    Both converge from the inside, so the estimate is a lower bound that has settled to `rtol`.
    Returns (kappa, trusted).  `trusted` is False when an LU reports a zero pivot / non-finite data, the
    iteration has not settled, or kappa lies within COND_GUARD of one of the reference's thresholds -- the
    caller then computes the exact value as the reference does."""
    n = A.shape[0]
    A = np.ascontiguousarray(A, dtype=np.complex128)
    rng = np.random.default_rng(20250607)              # private generator: the global streams are bookkeeping (appendix A)
    cA = _cabi.Context(device)
    cH = _cabi.Context(device)
    try:
        cA.set_matrix(A)
        cH.set_matrix(np.ascontiguousarray(A.conj().T))
        cA.pop_reserve(1)
        cH.pop_reserve(1)

        def unit():
            v = rng.standard_normal(n) + 1j * rng.standard_normal(n)
            return v / np.linalg.norm(v)

        cA.pop_put(POP_X, [0], unit()[None, :])
        smax, prev = 0.0, 0.0
        for it in range(80):
            smax = float(cA.svd_power_step([0])[0, 3])
            if not np.isfinite(smax) or smax == 0.0:
                return np.inf, False
            if it >= 4 and abs(smax - prev) <= 1e-3 * smax:
                break
            prev = smax
        zero, ps0 = np.zeros(1, dtype=np.complex128), np.zeros(1)
        v, smin, prev, settled = unit(), np.inf, np.inf, False
        for it in range(max_iter):
            cA.pop_put(POP_X, [0], v[None, :])
            if cA.shifted_lu_solve([0], zero, ps0, rhs_mode=0, pert_mode=PERT_NONE)[0] != 0:
                return np.inf, False
            w = cA.pop_get(POP_W, [0], n)[0]
            nw = np.linalg.norm(w)
            if not np.isfinite(nw) or nw == 0.0:
                return np.inf, False
            smin = 1.0 / nw
            cH.pop_put(POP_X, [0], (w / nw)[None, :])
            if cH.shifted_lu_solve([0], zero, ps0, rhs_mode=0, pert_mode=PERT_NONE)[0] != 0:
                return np.inf, False
            z = cH.pop_get(POP_W, [0], n)[0]
            nz = np.linalg.norm(z)
            if not np.isfinite(nz) or nz == 0.0:
                return np.inf, False
            v = z / nz
            if it >= 2 and abs(smin - prev) <= rtol * smin:
                settled = True
                break
            prev = smin
        kappa = smax / smin
        near = any(t / COND_GUARD <= kappa <= t * COND_GUARD for t in COND_THRESHOLDS)
        return kappa, bool(settled and np.isfinite(kappa) and not near)
    finally:
        cA.close()
        cH.close()


def singular_values_device(A, device=0):
    """All singular values of a dense matrix, descending, as the non-negative eigenvalues of the Hermitian embedding
    [[0, A], [A^H, 0]] (+-sigma_i and |rows - cols| zeros): tridiagonalised on the device (csrc/herm.hip), eigenvalues of the
    tridiagonal matrix by bisection there too (tridiagonal_eigenvalues).  The same absolute accuracy eps ||A|| as LAPACK's bidiagonal SVD -- what
    np.linalg.cond and scipy.linalg.svd(compute_uv=False) deliver -- at a fraction of their O(n^3) host time."""
    import scipy.linalg as sla
    A = np.asarray(A, dtype=np.complex128)
    r, c = A.shape
    Hm = np.zeros((r + c, r + c), dtype=np.complex128)
    Hm[:r, r:] = A
    Hm[r:, :r] = A.conj().T
    cH = _cabi.Context(device)
    try:
        cH.set_matrix(Hm)
        d, e = cH.herm_tridiag()
        w = tridiagonal_eigenvalues(cH, d, e)
    finally:
        cH.close()
    return np.maximum(np.sort(w)[::-1][: min(r, c)], 0.0)


def tridiagonal_eigenvalues(ctx, d, e):
    """Eigenvalues of the real symmetric tridiagonal (d, e), ascending: bisection on the device (csrc/herm.hip; within half an ulp
    of ||T|| of LAPACK dstebz) or, on MAUS_EIGH_TRIDIAG=host / non-finite input / a context without the entry point, LAPACK
    dsterf on the host."""
    import scipy.linalg as sla
    if d.shape[0] <= 1:
        return d.copy()
    if (os.environ.get("MAUS_EIGH_TRIDIAG", "auto") != "host" and hasattr(ctx, "herm_tridiag_eigvals")
            and np.isfinite(d).all() and np.isfinite(e).all()):
        return ctx.herm_tridiag_eigvals(d, e)
    return np.sort(sla.eigvalsh_tridiagonal(d, e))


class DeviceEngine:
    """One GPU context + slot allocator + the batched step."""

    _default = None

    def __init__(self, device: int = 0, pert_mode: str = "auto", gmres_compat: str = "rtol",
                 comm=None, ctx=None, eigh_mode: str = "auto"):
        self.ctx = ctx if ctx is not None else _cabi.Context(device)
        cap_blas_threads()
        # Hermitian eigendecomposition (AMS:161), once per matrix: 'host' = scipy.linalg.eigh, the reference's call, (lambda, V)
        # bit-identical to its; 'device' = reduction and back-transformation on the GPU, the tridiagonal eigenproblem by LAPACK
        # dstemr on the host (csrc/herm.hip); 'auto' = device from n = EIGH_DEVICE_MIN up.  MAUS_EIGH overrides.
        self.eigh_mode = os.environ.get("MAUS_EIGH", eigh_mode)
        self._owner = {}                        # id(candidate) -> rank that executes it this step
        self.pert_mode = pert_mode              # 'auto' | 'uniform' | 'mt19937' | 'none'
        self.gmres_compat = gmres_compat        # 'rtol' | 'scipy-legacy'  
        self.comm = comm                        # dist.PopulationComm or None
        if comm is not None and not getattr(comm, "on_device", False) and hasattr(self.ctx, "set_shared_device"):
            # host-side collectives (gloo): several ranks may be rehearsing on ONE GPU, so kernels whose workgroups wait
            # for each other (multi-workgroup LU panel) must not assume they have the device to themselves
            self.ctx.set_shared_device(True)
        if comm is not None and hasattr(comm, "attach"):
            comm.attach(self.ctx)               # 'rccl' transport: the communicator lives in the library, on this context's stream
        self._bound = None                      # matrix object currently on the device
        self._bound_b = None
        self._eig_cache = None                  # (matrix obj, evals) for the Hermitian shortcut
        self._typ_entry = None                  # (matrix obj, rms entry) for pert_matters
        self._free = []
        self._next_slot = 0
        self.store = CandidateStore()           # per-candidate bookkeeping, indexed by slot (candstore.py)
        self._deferred = None                   # candidates whose host vectors wait for one common transfer
        self.steps_executed = 0

    # ---- shared default engine (for stand-alone SolutionCandidate use) ----------------
    @classmethod
    def default(cls):
        if cls._default is None:
            cls._default = cls()
        return cls._default

    # ---- matrix / rhs binding -----------------------------------------------------------
    def bind_matrix(self, A):
        """The problem matrix on the device.  Sharded runs on RCCL: rank 0 alone stages it from the host, the others receive it
        device to device -- one collective, entered by every rank: a matrix that rank 0 bound on its own inside
        work it does for all (start-up diagnostics, comm.root_call) is broadcast from its device copy at the next bind."""
        comm = self.comm
        collective = comm is not None and comm.on_device and comm.world > 1 and not comm.in_root_call
        solo = getattr(self, "_bound_solo", False)
        if A is self._bound and not (collective and solo):
            return
        shape_changed = (self.ctx.rows, self.ctx.cols) != tuple(A.shape)
        if collective:
            self.ctx.comm_set_matrix(A, comm.rank, 0, resident_on_root=(comm.rank == 0 and solo and A is self._bound))
            comm.collectives += 1
            comm.bytes_gathered += int(A.nbytes)
            self._bound_solo = False
        else:
            self.ctx.set_matrix(A)
            self._bound_solo = comm is not None and comm.on_device and comm.world > 1
        self._bound = A
        if shape_changed:
            self._free = []
            self._next_slot = 0
            self._bound_b = None
            self.store = CandidateStore()       # the slots start again at 0: candidates of the old shape keep the old store

    def pert_matters(self, psi) -> np.ndarray:
        """AMS:49-52 adds 0.15*psi*((U1-.5)+i(U2-.5)) to H for GMRES as for the direct solve.  The device GMRES shares one
        A between all candidates and leaves the random term out while it is below the rounding of a matvec
        (|term| <= 0.075 psi < half an ulp of a typical entry of A); beyond that H_k is materialised (maus_gmres_pert)."""
        if self._typ_entry is None or self._typ_entry[0] is not self._bound:
            A = self._bound
            self._typ_entry = (A, float(np.linalg.norm(A)) / np.sqrt(max(1, A.size)))
        return 0.075 * np.abs(np.asarray(psi, dtype=np.float64)) >= 2.0 ** -53 * self._typ_entry[1]

    def bind_rhs(self, b):
        if b is None or b is self._bound_b:
            return
        self.ctx.set_rhs(b)
        self._bound_b = b

    # ---- slots ------------------------------------------------------------------------------
    def alloc_slot(self) -> int:
        if self._free:
            return self._free.pop()
        s = self._next_slot
        self._next_slot += 1
        if self._next_slot > self.ctx.pop_capacity():
            self.ctx.pop_reserve(max(64, 2 * self._next_slot))
        return s

    def free_slot(self, s: int) -> None:
        if s is not None and s >= 0:
            self._free.append(s)

    def attach(self, cand) -> None:
        """Give a candidate a device slot (and free it when the object dies)."""
        self.bind_matrix(cand.problem_matrix)
        cand._engine = self
        cand._ph.ctx = self.ctx
        cand._slot = self.alloc_slot()
        cand._st = self.store
        self.store.init_slot(cand._slot)
        self.store.records[cand._slot] = cand.__dict__.pop("_record_init", True)
        self.store.mat_tag[cand._slot] = self._mat_tag(cand.problem_matrix)
        cand._hist_seen = len(self.store.hist_log)
        weakref.finalize(cand, DeviceEngine._release, weakref.ref(self), cand._slot, id(self._bound))

    def _mat_tag(self, M) -> int:
        """Small integer per matrix OBJECT candidates were constructed on (the residual of AMS:295-301 is taken against that
        object)."""
        tags = self.__dict__.setdefault("_mat_tags", {})
        t = tags.get(id(M))
        if t is None or t[1]() is not M:                  # (an id can come back after its object has died)
            self._mat_tag_next = getattr(self, "_mat_tag_next", 0) + 1
            t = tags[id(M)] = (self._mat_tag_next, weakref.ref(M))
        return t[0]

    @staticmethod
    def _release(eng_ref, slot, mat_id):
        eng = eng_ref()
        if eng is not None and eng._bound is not None and id(eng._bound) == mat_id:
            eng.free_slot(slot)

    @staticmethod
    def _slots(cands) -> np.ndarray:
        return np.fromiter(map(_SLOT_OF, cands), dtype=np.int64, count=len(cands))

    def begin_deferred_push(self) -> None:
        self._deferred = []

    def end_deferred_push(self) -> None:
        """One pop_put per population array for the candidates constructed since begin_deferred_push (the spawns of
        AMS:533-549: up to 15 per loop body, one 22 us call each before)."""
        cands, self._deferred = self._deferred, None
        if not cands:
            return
        from .solver import ProblemType
        cands = list({id(c): c for c in cands}.values())    # a candidate re-initialised twice is pushed once, with its last vectors
        slots = self._slots(cands)
        hv = [c for c in cands if c._hv is not None]
        if hv:
            self.ctx.pop_put(POP_X, self._slots(hv), np.stack([c._hv for c in hv]))
        hu = [c for c in cands if c._hu is not None and c.problem_type == ProblemType.SVD]
        if hu:
            self.ctx.pop_put(POP_U, self._slots(hu), np.stack([c._hu for c in hu]))
        self.store.dev_valid[slots] = True

    def _bulk_pull(self, cands, slots=None) -> None:
        """Refresh stale host mirrors of many candidates with one transfer per array."""
        from .solver import ProblemType
        if not cands:
            return
        st = self.store
        if slots is None:
            slots = self._slots(cands)
        ix = np.nonzero(~st.host_valid[slots])[0]
        if not ix.size:
            return
        stale = [cands[k] for k in ix.tolist()]
        sl = slots[ix]
        X = self.ctx.pop_get(POP_X, sl, stale[0]._len_v())
        U = None
        if stale[0].problem_type == ProblemType.SVD:
            U = self.ctx.pop_get(POP_U, sl, stale[0].M_rows)
        for k, c in enumerate(stale):
            c._hv = X[k]
            if U is not None:
                c._hu = U[k]
        st.host_valid[sl] = True

    def _log_history(self, cands, slots) -> None:
        """AMS:303-304 for every stepped candidate: param_history.append(get_current_solution_params()),
        residual_history.append(residual_k).  Small problems (n <= 512): the host mirrors are refreshed with one transfer and
        every candidate appends its tuple.  Above that the vectors are appended device-to-device to the history store (SURVEY
        f-4: device-backed param_history, pulled on access) and the step leaves ONE record in the store's log -- the
        candidates' lists catch up when somebody reads them (SolutionCandidate._replay_history), so a loop body does not
        touch 6 144 Python objects to remember what it did."""
        from .solver import ProblemType
        c0 = cands[0]
        st = self.store
        if not c0._lazy_hist:
            ix = np.nonzero(st.records[slots])[0]
            if ix.size:
                self._bulk_pull(cands if ix.size == len(cands) else [cands[k] for k in ix.tolist()], slots[ix])
            for c in cands:
                c._record_history()
            return
        recording = st.records[slots]
        href = None
        hidx = None
        if recording.any():
            rs = slots if recording.all() else slots[recording]
            svd = c0.problem_type == ProblemType.SVD
            lv = c0._len_v()
            iv = self.ctx.hist_append(POP_X, rs, lv)
            iu = self.ctx.hist_append(POP_U, rs, c0.M_rows) if svd else None
            gen = self.ctx.hist_generation() if hasattr(self.ctx, "hist_generation") else 0      # once per step, not per candidate
            href = (gen, iu, c0.M_rows, iv, lv) if svd else (gen, iv, lv)
            hidx = np.where(recording, np.cumsum(recording) - 1, -1)
        kind = c0.problem_type
        if kind == ProblemType.EIGENVALUE:
            scal, cplx = (st.lam[slots], st.lam_kind[slots], st.lam_obj[slots]), True
        elif kind == ProblemType.SVD:
            scal, cplx = (st.sig[slots], st.sig_kind[slots], st.sig_obj[slots]), False
        else:
            scal, cplx = None, False
        st.hist_log.append(HistoryRecord(slots.copy(), (st.res[slots], st.res_kind[slots], st.res_obj[slots]), scal, cplx, hidx, href))

    # ---- perturbation mode ---------------------------------------------------------------
    def _pert(self, n: int) -> int:
        if self.pert_mode == "uniform":
            return PERT_UNIFORM
        if self.pert_mode == "none":
            return PERT_NONE
        if self.pert_mode == "mt19937":
            return PERT_MT19937
        # auto: small problems upload the exact host draws; large ones regenerate the same draws on the
        # device from the NumPy state (bit-identical H either way)
        return PERT_UNIFORM if n <= 256 else PERT_MT19937

    # ======================================================================================
    # device phases, sharded over ranks when a communicator is present (dist.py)
    # ======================================================================================
    def _mine(self, cands):
        if self.comm is None:
            return None
        r = self.comm.rank
        return [k for k, c in enumerate(cands) if self._owner.get(id(c), 0) == r]

    def _exchange(self, cands, local: np.ndarray) -> np.ndarray:
        """All-gather per-candidate rows (float64) computed by their owners into list order."""
        comm = self.comm
        own = np.array([self._owner.get(id(c), 0) for c in cands], dtype=np.int64)
        counts = [int(np.sum(own == r)) for r in range(comm.world)]
        full = comm.allgather_rows(np.ascontiguousarray(local, dtype=np.float64), counts)
        order = np.concatenate([np.nonzero(own == r)[0] for r in range(comm.world)]) if len(cands) else np.zeros(0, int)
        out = np.empty_like(full)
        out[order] = full
        return out

    def d_rayleigh(self, cands):
        mine = self._mine(cands)
        if mine is None:
            return self.ctx.matvec_rayleigh([c._slot for c in cands])
        loc = np.zeros((len(mine), 4))
        if mine:
            num, den = self.ctx.matvec_rayleigh([cands[k]._slot for k in mine])
            loc = np.column_stack([num.real, num.imag, den.real, den.imag])
        f = self._exchange(cands, loc)
        return f[:, 0] + 1j * f[:, 1], f[:, 2] + 1j * f[:, 3]

    def d_lu_solve(self, cands, shift, psi, rhs_mode, pert, pert_data):
        mine = self._mine(cands)
        if mine is None:
            return self.ctx.shifted_lu_solve([c._slot for c in cands], shift, psi, rhs_mode=rhs_mode,
                                             pert_mode=pert, pert_data=pert_data)
        loc = np.zeros((len(mine), 1))
        if mine:
            if pert == PERT_MT19937:
                sub = (pert_data[0], pert_data[1], pert_data[2], np.asarray(pert_data[3])[mine])
            else:
                sub = None if pert_data is None else pert_data[mine]
            st = self.ctx.shifted_lu_solve([cands[k]._slot for k in mine], shift[mine], psi[mine], rhs_mode=rhs_mode,
                                           pert_mode=pert, pert_data=sub)
            loc = st.astype(np.float64)[:, None]
        return self._exchange(cands, loc)[:, 0].astype(np.int32)

    def d_gmres(self, cands, shift, psi, rhs_mode, use_j):
        mine = self._mine(cands)
        if mine is None:
            return self.ctx.gmres([c._slot for c in cands], shift, psi, rhs_mode, use_j)
        loc = np.zeros((len(mine), 3))
        if mine:
            info, inner, status = self.ctx.gmres([cands[k]._slot for k in mine], shift[mine], psi[mine], rhs_mode, use_j[mine])
            loc = np.column_stack([info, inner, status]).astype(np.float64)
        f = self._exchange(cands, loc).astype(np.int32)
        return f[:, 0], f[:, 1], f[:, 2]

    def d_gmres_pert(self, cands, shift, psi, rhs_mode, want_j, pert, pert_data):
        mine = self._mine(cands)
        if mine is None:
            info, inner, status, _ = self.ctx.gmres_pert([c._slot for c in cands], shift, psi, rhs_mode, want_j, pert, pert_data)
            return info, inner, status
        loc = np.zeros((len(mine), 3))
        if mine:
            if pert == PERT_MT19937:
                sub = (pert_data[0], pert_data[1], pert_data[2], np.asarray(pert_data[3])[mine])
            else:
                sub = None if pert_data is None else pert_data[mine]
            info, inner, status, _ = self.ctx.gmres_pert([cands[k]._slot for k in mine], shift[mine], psi[mine], rhs_mode,
                                                         want_j[mine], pert, sub)
            loc = np.column_stack([info, inner, status]).astype(np.float64)
        f = self._exchange(cands, loc).astype(np.int32)
        return f[:, 0], f[:, 1], f[:, 2]

    def d_relax(self, cands, alpha, normalise):
        mine = self._mine(cands)
        if mine is None:
            return self.ctx.relax_normalise([c._slot for c in cands], alpha, normalise=normalise)
        loc = np.zeros((len(mine), 1))
        if mine:
            loc = self.ctx.relax_normalise([cands[k]._slot for k in mine], alpha[mine], normalise=normalise)[:, None]
        return self._exchange(cands, loc)[:, 0]

    def d_residual(self, kind, cands, lam, slots=None):
        mine = self._mine(cands)
        if mine is None:
            return self.ctx.residual(kind, self._slots(cands) if slots is None else slots, lam)
        loc = np.zeros((len(mine), 2))
        if mine:
            res, fin = self.ctx.residual(kind, [cands[k]._slot for k in mine], None if lam is None else lam[mine])
            loc = np.column_stack([res, fin.astype(np.float64)])
        f = self._exchange(cands, loc)
        return f[:, 0], f[:, 1] != 0

    def d_svd_power(self, cands, slots=None):
        """Speculative power step: norms only, the candidates' vectors stay as they are (d_svd_commit applies it)."""
        mine = self._mine(cands)
        if mine is None:
            return self.ctx.svd_power_propose(self._slots(cands) if slots is None else slots)
        loc = np.zeros((len(mine), 4))
        if mine:
            loc = self.ctx.svd_power_propose([cands[k]._slot for k in mine])
        return self._exchange(cands, loc)

    def d_svd_commit(self, cands, slots=None):
        mine = self._mine(cands)
        if mine is None:
            if cands:
                self.ctx.svd_commit(self._slots(cands) if slots is None else slots)
            return
        own = [cands[k] for k in mine]
        if own:
            self.ctx.svd_commit([c._slot for c in own])

    def d_herm_match(self, cands):
        mine = self._mine(cands)
        if mine is None:
            return self.ctx.herm_match([c._slot for c in cands])
        loc = np.zeros((len(mine), 2))
        if mine:
            idx, nrm = self.ctx.herm_match([cands[k]._slot for k in mine])
            loc = np.column_stack([idx.astype(np.float64), nrm])
        f = self._exchange(cands, loc)
        return f[:, 0].astype(np.int32), f[:, 1]

    def d_gram(self, cands, which, length):
        """abs-free Gram block G[i, j] = vdot(x_i, x_j) of the candidates' device rows (SURVEY f-2).  Every rank
        holds every row after _sync_rows, so the block is computed redundantly (and identically) on each rank."""
        return self.ctx.gram(which, [c._slot for c in cands], length)

    def _sync_rows(self, cands) -> None:
        """After a sharded step: every rank receives the rows its peers updated."""
        if self.comm is None or not cands:
            return
        from .solver import ProblemType
        mine = self._mine(cands)
        arrays = [(POP_X, cands[0]._len_v())]
        if cands[0].problem_type == ProblemType.SVD:
            arrays.append((POP_U, cands[0].M_rows))
        if self.comm.on_device and os.environ.get("MAUS_DIST_DEVICE_ROWS", "1") != "0":
            # RCCL straight between the population arrays (dist.PopulationComm.sync_rows_device)
            by_rank = [[c._slot for c in cands if self._owner.get(id(c), 0) == r] for r in range(self.comm.world)]
            for which, length in arrays:
                self.comm.sync_rows_device(self.ctx, which, by_rank, length)
            return
        mine_set = set(mine)
        others = [k for k in range(len(cands)) if k not in mine_set]
        for which, length in arrays:
            loc = np.zeros((len(mine), 2 * length))
            if mine:
                loc = self.ctx.pop_get(which, [cands[k]._slot for k in mine], length).view(np.float64)
            full = self._exchange(cands, loc)
            if others:
                rows = np.ascontiguousarray(full[others]).view(np.complex128)
                self.ctx.pop_put(which, [cands[k]._slot for k in others], rows)

    # ======================================================================================
    # the step
    # ======================================================================================
    def step(self, cands, A, b, strat, know, slots=None) -> None:
        """update_solution_step for every candidate of `cands` (list order = RNG order)."""
        if not cands:
            return
        from .solver import ProblemType, SolutionCandidate
        S = SolutionCandidate.State
        kind = cands[0].problem_type
        self.bind_matrix(A)
        self.bind_rhs(b)
        n_vec = max(cands[0].M_rows, cands[0].M_cols)
        # host mirrors double as the pre-step backup wherever a speculative batch may have to be undone
        # (the SVD step needs none: its speculative pass is only committed for the candidates in front of an event)
        need_backup = kind != ProblemType.SVD and self._pert(cands[0].N_diag) == PERT_UNIFORM
        if self.comm is not None:
            own = self.comm.owners(len(cands))
            self._owner = {id(c): int(own[k]) for k, c in enumerate(cands)}
        st = self.store
        if slots is None or len(slots) != len(cands):
            slots = self._slots(cands)
        if need_backup:
            self._bulk_pull(cands, slots)
        st.set_all("b_obj", slots, b)                       # AMS:146
        st.copy_real("prev", "res", slots)                  # AMS:147
        for k in np.nonzero(~st.dev_valid[slots])[0].tolist():
            cands[k]._push()                                # make sure device rows are current
        self.steps_executed += len(cands)

        todo = list(cands)
        if kind == ProblemType.EIGENVALUE and know.get("is_hermitian", False):   # AMS:155
            todo = self._hermitian(todo, A, slots)
            if todo:
                slots = self._slots(todo)
        if todo:
            if kind == ProblemType.SVD:
                self._svd(todo, A, strat, slots)
            else:
                self._solve(todo, A, b, strat, know)
            self._sync_rows(todo)                       # peers' updated rows, before anything reads them back
            self._finish(todo, A, b, strat, slots)

    # ---- Hermitian shortcut (AMS:155-221) ----------------------------------------------
    def seed_eigh(self, A, evals, evecs):
        """eigh(A) computed by the caller (the start-up diagnostics took the condition number from it): use it as the
        once-per-matrix decomposition of the shortcut."""
        if evecs is not None:                   # None: device_eigh left V resident on the device
            self.ctx.set_eigvecs(evecs)
        self._eig_cache = (A, evals)

    def use_device_eigh(self, n: int) -> bool:
        if not hasattr(self.ctx, "herm_tridiag") or self.eigh_mode == "host":
            return False
        return self.eigh_mode == "device" or n >= EIGH_DEVICE_MIN

    def device_eigh(self, A):
        """scipy.linalg.eigh(A) (LAPACK zheevr = zhetrd + dstemr + zunmtr) with zhetrd and zunmtr on the device: eigenvalues
        on the host, the eigenvector matrix resident on the device as if uploaded with set_eigvecs.  Same reflector
        conventions as LAPACK (first row of V real); the sign of a column is dstemr's for our T and differs from the
        reference's in about one column out of ten -- as it does between LAPACK's own T and that T moved by one ulp.
        73 s -> seconds at n = 8192."""
        import scipy.linalg as sla
        self.bind_matrix(A)
        d, e = self.ctx.herm_tridiag()
        self.tridiag_de = (d, e)               # the real tridiagonal T (diagnostics and tests)
        n = A.shape[0]
        if n == 1:
            self.ctx.herm_backtransform(np.ones((1, 1)))
            return d.copy()
        # The tridiagonal eigenproblem: on the device (bisection + twisted factorisation, csrc/herm.hip) when its own checks
        # say the spectrum is well separated and the residuals are at rounding level -- a vector is then off by
        # eps ||T|| / gap <= ~1e-9 -- and by LAPACK dstemr on the host (the kernel zheevr uses; 7.4 of the decomposition's
        # 8.5 s at n = 8192) for clustered spectra or on MAUS_EIGH_TRIDIAG=host.
        self.tridiag_solver = "host"
        if os.environ.get("MAUS_EIGH_TRIDIAG", "auto") != "host" and hasattr(self.ctx, "herm_tridiag_eig") and np.isfinite(d).all() and np.isfinite(e).all():
            evals, (gap, resid, tnorm) = self.ctx.herm_tridiag_eig(d, e)
            self.tridiag_diag = (gap, resid, tnorm)
            if np.isfinite(evals).all() and gap >= TRIDIAG_MIN_GAP and resid <= TRIDIAG_MAX_RESID:
                self.ctx.herm_backtransform(None)
                self.tridiag_solver = "device"
                return evals
        evals, Z = sla.eigh_tridiagonal(d, e)                      # LAPACK dstemr
        self.ctx.herm_backtransform(Z)
        return evals

    def _eigh_once(self, A):
        """One decomposition per matrix version instead of one per candidate; the same LAPACK call as the
        reference (AMS:161), so (lambda, V) are bit-identical to its.  In a sharded run rank 0 alone decomposes -- with
        the node's BLAS threads, the other ranks are waiting -- and broadcasts the eigenvalues (host) and the
        eigenvector matrix (device to device over RCCL): N ranks must not run N copies of a 73-second eigh (n = 8192) on
        one thread each.  Returns (eigenvalues, None) with V resident on the device, or (None, error text)."""
        import scipy.linalg as sla
        comm = self.comm
        n = A.shape[0]
        dev = self.use_device_eigh(n)
        if comm is None or comm.world == 1:
            try:
                if dev:
                    return self.device_eigh(A), None
                evals, evecs = sla.eigh(A)
            except np.linalg.LinAlgError as e:
                return None, str(e)
            self.ctx.set_eigvecs(evecs)
            return evals, None
        held = {}

        def decompose():
            # LinAlgError is the reference's own failure mode (AMS:180: the candidates fall back to the iteration) and travels
            # as a result; anything else that stops rank 0 here is raised on every rank by root_call
            try:
                with comm.all_blas_threads():
                    if dev:
                        return "", self.device_eigh(A)             # V stays on rank 0's device; broadcast from there
                    ev, held["evecs"] = sla.eigh(A)
                    return "", ev
            except np.linalg.LinAlgError as e:
                return str(e) or "eigh failed", None
        err, ev = comm.root_call(decompose)
        if err:
            return None, err
        evals, evecs = np.ascontiguousarray(ev, dtype=np.float64), held.get("evecs")
        comm.bcast_eigvecs(self.ctx, evecs, n)
        return evals, None

    def seed_eigh_distributed(self, A, evals, evecs):
        """seed_eigh for a sharded run: rank 0 passes the decomposition its start-up diagnostics computed, the other ranks
        pass None and receive it."""
        comm = self.comm
        n = A.shape[0]
        ev = np.ascontiguousarray(evals, dtype=np.float64) if comm.rank == 0 else np.empty(n, dtype=np.float64)
        comm.bcast_array(ev)
        comm.bcast_eigvecs(self.ctx, evecs, n)
        self._eig_cache = (A, ev)

    def _hermitian(self, cands, A, slots):
        from .solver import SolutionCandidate
        S = SolutionCandidate.State
        if self._eig_cache is not None and self._eig_cache[0] is not A and self._eig_cache[0].shape == A.shape \
                and np.array_equal(self._eig_cache[0], A):
            self._eig_cache = (A, self._eig_cache[1])     # another solver on the same engine with an equal matrix: V is still resident
        if self._eig_cache is None or self._eig_cache[0] is not A:
            evals, err = self._eigh_once(A)
            if evals is None:
                for c in cands:
                    print(f"Candidate {c.id}: Dense Hermitian solver (eigh) failed: {err}. Falling back.")
                return cands
            self._eig_cache = (A, evals)
        evals = self._eig_cache[1]
        idx, _ = self.d_herm_match(cands)
        lam = evals[idx]
        res, _fin = self.d_residual(KIND_EIG, cands, lam.astype(np.complex128))
        st = self.store
        st.host_valid[slots] = False                       # _invalidate(): the device rows are the new state
        st.dev_valid[slots] = True
        self._sync_rows(cands)
        # lambda_k = evals[idx] keeps the dtype of the eigenvalue array (np.float64 from eigh): assigned as objects
        for k, c in enumerate(cands):
            c.lambda_k = lam[k]
        st.set_real("res", slots, res, F_NP)
        st.state[slots] = S.CONVERGED.value
        st.stuck[slots] = 0
        st.retries[slots] = 0
        st.set_real("w", slots, 1.0, F_PY)
        self._log_history(cands, slots)
        return []

    # ---- SVD alternating power step (AMS:227-255) -----------------------------------------
    def _svd(self, cands, A, strat, slots):
        from .solver import SolutionCandidate
        S = SolutionCandidate.State
        st = self.store
        tiny = SIGMA_SIMILARITY_TOL_ABS / 100
        i = 0
        while i < len(cands):
            run = cands[i:]
            rslots = slots[i:]
            # speculative: the whole run; norms = (||v_in||, sigma1, ||u||, sigma2).  Nothing is written to the candidates'
            # vectors until the commit below, so the candidates behind an event need no restoring
            norms = self.d_svd_power(run, rslots)
            # first candidate that takes an exceptional branch
            bad = (norms[:, 0] < 1e-10) | (norms[:, 2] < 1e-10) | ~np.isfinite(norms).all(axis=1)
            ev = int(np.argmax(bad)) if bad.any() else None
            good = run if ev is None else run[:ev]
            ng = len(good)
            gs = rslots[:ng]
            self.d_svd_commit(good, gs)
            sig = np.where(norms[:ng, 3] > norms[:ng, 1], norms[:ng, 3], norms[:ng, 1])   # max(s1, s2), AMS:234, 241 (finite here)
            st.set_real("sig", gs, sig, F_NP)
            st.host_valid[gs] = False                                     # _invalidate(): the device rows are the new state
            st.dev_valid[gs] = True
            conv = sig < tiny
            for k in np.nonzero(conv)[0].tolist():                        # AMS:243-247
                c = good[k]
                c.residual_k = strat.get("current_convergence_threshold", 1e-6) * 0.1
                c.state = S.CONVERGED
                c.stuck_counter = 0
                # AMS:246-247: a collapsed u_k / right_v_k is replaced by ones/sqrt(dim).  ||u_k|| is norms[k, 2]
                # (a collapse there has already left through the exception path, like AMS:236-239 raises before
                # these lines); right_v_k = s / (sigma2 if sigma2 > 1e-10 else 1), so its norm is below 1e-10
                # exactly when sigma2 is.  Reachable only at a rounding knife edge.
                if norms[k, 2] < 1e-10:
                    c.u_k = np.ones(c.M_rows, dtype=np.complex128) / np.sqrt(c.M_rows)
                    c._push(force=True)
                if norms[k, 3] < 1e-10:
                    c.right_v_k = np.ones(c.M_cols, dtype=np.complex128) / np.sqrt(c.M_cols)
                    c._push(force=True)
            dec = gs[~conv] if conv.any() else gs
            st.stuck[dec] = np.maximum(st.stuck[dec] - 1, 0)              # max(0, stuck - 1), AMS:248
            if ev is None:
                break
            self._svd_exception(run[ev], norms[ev])
            i += ev + 1

    def _svd_exception(self, c, nrm):
        """Sequential semantics for the collapse / failure branches (AMS:229-232, 236-239, 249-255)."""
        from .solver import SolutionCandidate
        S = SolutionCandidate.State
        # (the device rows of c are still its pre-step vectors: the speculative pass was not committed for it)
        if nrm[0] < 1e-10:                                   # right vector collapsed (AMS:229-232)
            v = (np.random.rand(c.M_cols) + 1j * np.random.rand(c.M_cols))
            v /= np.linalg.norm(v)
            c.stuck_counter += 1
            c.num_resets += 1
        elif nrm[2] < 1e-10:                                 # left vector collapsed (AMS:236-239)
            c.sigma_k = np.float64(nrm[1])
            u = (np.random.rand(c.M_rows) + 1j * np.random.rand(c.M_rows))
            u /= np.linalg.norm(u)
            c.stuck_counter += 1
            c.num_resets += 1
        # failure ladder (AMS:249-255); also reached for non-finite norms?  No: the reference only
        # raises on the two collapses, non-finite values flow through.  Handle that case as success.
        if nrm[0] < 1e-10 or nrm[2] < 1e-10:
            c.stuck_counter += 1
            c.w_k *= 0.001
            c.alpha_local_step *= 0.5
            c.state = S.STUCK
            if c.stuck_counter >= MAX_STUCK_FOR_RETIREMENT:
                c.state = S.RETIRED
            c.u_k = (np.random.rand(c.M_rows) + 1j * np.random.rand(c.M_rows)) / np.sqrt(c.M_rows)
            c.right_v_k = (np.random.rand(c.M_cols) + 1j * np.random.rand(c.M_cols)) / np.sqrt(c.M_cols)
            c.sigma_k = 1.0
            c._push(force=True)
        else:
            # non-finite norms: redo this candidate alone and accept what comes out
            norms = self.d_svd_power([c])
            self.d_svd_commit([c])
            c.sigma_k = max(np.float64(norms[0, 1]), np.float64(norms[0, 3]))
            c._invalidate()
            c.stuck_counter = max(0, c.stuck_counter - 1)

    # ---- eig / linear: Rayleigh, shifted solve, relaxed update (AMS:256-293) ---------------
    def _solve(self, cands, A, b, strat, know):
        from .solver import ProblemType
        kind = cands[0].problem_type
        is_eig = kind == ProblemType.EIGENVALUE
        n = cands[0].N_diag
        aggr = strat.get("overall_psi_aggression_factor", 1.0)
        max_retries = strat.get("max_psi_retries", MAX_PSI_ATTEMPTS)
        pref = know.get("local_solver_preference", DIRECT)
        base_psi = PSI_EPSILON_BASE * aggr                              # AMS:224
        pert = self._pert(n)
        words = 4 * n * n                                               # MT19937 words per dense attempt
        # LU workspace sized once for this rank's share of the population plus the growth of the next ~20 iterations
        # (<= 15 spawns each whatever the population, AMS:533-534): it must not be re-allocated inside somebody's timed
        # step (freeing and mapping ~100 GB takes seconds; HBM is otherwise idle)
        world = self.comm.world if self.comm is not None else 1
        share = -(-len(cands) // world)
        if pref == DIRECT or self.gmres_compat == "scipy-legacy":
            self.ctx.lu_reserve(n, max(2 * share, share + -(-320 // world)))
        # (GMRES preferred: the LU only serves the candidates whose GMRES attempt fails, and its workspace is sized for them
        # when that happens -- reserving the whole population's H_k here cost configs[2] 4.7 s in its first loop body and a
        # second 270 GB allocation in the next one, for a workspace the step never touched)

        i = 0
        self._pre_resid = {}
        while i < len(cands):
            run = cands[i:]
            slots = self._slots(run)
            st = self.store
            # Sharded run, direct solver, stream-independent host side ('mt19937' / 'none'): this rank's share of the WHOLE step --
            # Rayleigh quotient, shifted solve, relaxed update, residual -- before anything is exchanged, then one 64-byte record
            # per candidate.  Accepted if no candidate of the run takes an exceptional branch (the common case by far);
            # otherwise undone and the phase-by-phase path below takes the run.
            if (self.comm is not None and pert != PERT_UNIFORM and pref == DIRECT
                    and self._share_step(run, A, is_eig, base_psi, pert, words, n)):
                i += len(run)
                continue
            if is_eig:
                num, den = self.d_rayleigh(run)
                vnorm = np.sqrt(den.real)
                collapsed = np.nonzero(vnorm < 1e-10)[0]
                if collapsed.size and collapsed[0] == 0:                # E2 on the head of the run
                    c = run[0]
                    v = (np.random.rand(n) + 1j * np.random.rand(n))
                    v /= np.linalg.norm(v)
                    c.v_k = v
                    c.stuck_counter += 1
                    c.num_resets += 1
                    print(f"Candidate {c.id}: Eigenvector collapsed, reinitialized randomly before InverseIterateSolver.")
                    c._push(force=True)
                    continue
                if collapsed.size:
                    k = int(collapsed[0])
                    run, slots, num, den = run[:k], slots[:k], num[:k], den[:k]
                lam = np.empty(len(run), dtype=np.complex128)
                for k in range(len(run)):                               # AMS:264-268
                    if np.abs(den[k]) < 1e-12:
                        lam[k] = complex(0.0, 0.0)
                    else:
                        lam[k] = num[k] / den[k]
                st.lam[slots] = lam                                     # lambda_k = the np.complex128 quotient
                st.lam_kind[slots] = C_NP
                st.lam_obj[slots] = MISSING
                shift = lam
            else:
                shift = np.zeros(len(run), dtype=np.complex128)

            # --- first attempt of every candidate, batched (attempt 0, preferred method) ---
            stuck = st.stuck[slots]
            psi0 = np.array([(base_psi * (10 ** (0 / 2.0)) * (10 ** (s / 3.0))).real for s in stuck])   # AMS:44
            first_method = pref
            rng_after = None
            pert_data = None
            rng_start = np.random.get_state()
            legacy_gmres = (first_method == GMRES and self.gmres_compat == "scipy-legacy")
            if pert == PERT_UNIFORM:
                pert_data = np.empty((len(run), 2, n, n))
                rng_after = []
                for k in range(len(run)):
                    if legacy_gmres:                                    # the swallowed TypeError attempt draws too
                        np.random.rand(n, n); np.random.rand(n, n)
                    pert_data[k, 0] = np.random.rand(n, n)
                    pert_data[k, 1] = np.random.rand(n, n)
                    rng_after.append(np.random.get_state())
            per_cand_words = words * (2 if legacy_gmres else 1)
            if pert == PERT_MT19937:
                # candidate k of the run uses the 4N^2 words at lead + k*per_cand_words of the current stream
                pert_data = (rng_start, per_cand_words, per_cand_words - words, np.arange(len(run), dtype=np.int32))
            ahead = None
            fb = np.zeros(len(run), dtype=bool)         # candidates that took the batched direct-solver retry
            if pert != PERT_UNIFORM and len(run) >= 8:
                ahead = _AsyncStreamAdvance(per_cand_words * len(run))       # overlaps the GPU batch below
            if first_method == DIRECT or legacy_gmres:
                status = self.d_lu_solve(run, shift, psi0, 0 if is_eig else 1, pert, pert_data)
                ok = status == 0
            else:
                ok = self._gmres_batch(run, shift, psi0, stuck, is_eig, pert, pert_data)
                # escalated psi: the GMRES iterates depend on the draws (dense mode), so a failed attempt -- whose retry
                # moves the stream position of everybody behind it -- ends the run like any other RNG event
                dense_any = pert != PERT_NONE and bool(self.pert_matters(psi0).any())
                if pert != PERT_UNIFORM and not ok.all() and not dense_any:
                    # AMS:99-103 in batch: a failed attempt 0 with the preferred (GMRES) method is retried with the
                    # direct solver at the same attempt index.  For ill-conditioned ("Fragile") systems that is the
                    # common case, not the exception -- one batched LU instead of one ladder per candidate.  The
                    # retry draws its own rand(N,N) pair (E3), right after the pair of the candidate's GMRES attempt:
                    # candidate k's first attempt sits at attempt slot k + (number of earlier fallbacks).
                    F = np.nonzero(~ok)[0]
                    slot0 = np.concatenate([[0], np.cumsum(1 + (~ok).astype(np.int64))])[:-1]
                    pd = None
                    if pert == PERT_MT19937:
                        pd = (rng_start, words, 0, (slot0[F] + 1).astype(np.int32))
                    st_lu = self.d_lu_solve([run[k] for k in F], shift[F], psi0[F], 0 if is_eig else 1, pert, pd)
                    fb[F] = True
                    ok = ok.copy()
                    ok[F] = st_lu == 0
            bad = np.nonzero(~ok)[0]
            nb = int(bad[0]) if bad.size else len(run)

            # --- relaxed update of the good prefix (AMS:280-286) ---
            e4 = None
            if nb > 0:
                if (st.alpha_kind[slots[:nb]] == EXACT).any():          # an alpha of a type some caller assigned
                    alpha = np.array([complex(c.alpha_local_step) for c in run[:nb]], dtype=np.complex128)
                else:
                    alpha = st.alpha[slots[:nb]].astype(np.complex128)
                if is_eig and pert == PERT_MT19937:
                    # device-side snapshot of the run's vectors (POP_U is unused by eig problems): an E4 event moves the
                    # draws of everybody behind it, and their speculative update must then be undone
                    self.ctx.pop_copy(POP_U, POP_X, slots[:nb])
                nrm = self.d_relax(run[:nb], alpha, is_eig)
                if is_eig:
                    tiny = np.nonzero(~(nrm > 1e-10))[0]
                    if tiny.size:
                        e4 = int(tiny[0])
            # E4 (AMS:283) consumes 2 x rand(N): with a stream-dependent perturbation (host draws or device-regenerated
            # draws alike) the candidates behind it are recomputed from their new stream positions; with the perturbation
            # dropped ('none') their results do not depend on the stream and are kept
            nvalid = nb if (e4 is None or pert == PERT_NONE) else e4 + 1
            clean = ahead is not None and nvalid == len(run) and e4 is None and not fb.any()
            if ahead is not None and not clean:
                ahead.discard()
            # --- RNG replay + bookkeeping for the accepted candidates, in list order ---
            if pert == PERT_UNIFORM:
                np.random.set_state(rng_after[nvalid - 1] if nvalid > 0 else rng_start)
            pending_words = 0          # E3 consumption not yet applied to the stream (one jump per run)
            vs = slots[:nvalid]
            st.retries[vs] = 0                                          # attempts == 0 (AMS:278)
            st.host_valid[vs] = False                                   # _invalidate(): the device rows are the new state
            st.dev_valid[vs] = True
            if is_eig and e4 is not None:
                for k in range(nvalid):
                    c = run[k]
                    if pert != PERT_UNIFORM and not clean:
                        pending_words += per_cand_words * (2 if fb[k] else 1)   # E3 (attempt 0, and its direct-solver retry)
                    if k == e4 or (pert != PERT_UNIFORM and not (nrm[k] > 1e-10)):
                        _advance_numpy_stream(pending_words)
                        pending_words = 0
                        c.v_k = (np.random.rand(n) + 1j * np.random.rand(n)) / np.sqrt(n)     # E4 (AMS:283)
                        c._push(force=True)
            elif pert != PERT_UNIFORM and not clean:
                pending_words = per_cand_words * (nvalid + int(fb[:nvalid].sum()))      # E3 of every accepted candidate
            st.stuck[vs] = np.maximum(st.stuck[vs] - 1, 0)              # AMS:286
            if clean:
                ahead.apply()                                           # whole run's E3 consumption, precomputed
            else:
                _advance_numpy_stream(pending_words)
            if pert == PERT_UNIFORM and nvalid < nb:
                for c in run[nvalid:nb]:                                # speculative relax undone
                    c._restore_device()
            if pert == PERT_MT19937 and nvalid < nb:
                self.ctx.pop_copy(POP_X, POP_U, slots[nvalid:nb])
            i += nvalid
            if nvalid == nb and nb < len(run):
                # --- the candidate whose first attempt failed: exact sequential ladder ---
                c = run[nb]
                if pert == PERT_UNIFORM:
                    np.random.set_state(rng_after[nb])
                else:
                    _advance_numpy_stream(per_cand_words * (2 if fb[nb] else 1))
                failed_method = DIRECT if (first_method == DIRECT or legacy_gmres or fb[nb]) else GMRES
                self._ladder(c, shift[nb], base_psi, max_retries, pref, failed_method, is_eig, n, pert)
                if pert == PERT_UNIFORM:
                    for cc in run[nb + 1:]:
                        cc._restore_device()
                i += 1

    def _share_step(self, run, A, is_eig, base_psi, pert, words, n) -> bool:
        """one record per candidate and step.  Every rank runs AMS:264-286 + 295-301 for the candidates it owns,
        speculatively (attempt 0 succeeds, no E2 / E4 event), and the owners' results travel in ONE all-gather of
        [num.re, num.im, den.re, den.im, status, norm, residual, finite] (64 bytes).  Every rank then takes the same decision
        from the same table: accept the run -- bookkeeping as in the phase-by-phase path, two collectives per loop body
        with the row exchange -- or, on any exceptional branch, restore the vectors from the device-side snapshot and
        return False."""
        if any(c.problem_matrix is not A for c in run):             # residuals against another matrix object
            return False
        mine = self._mine(run)
        loc = np.zeros((len(mine), 8))
        ahead = _AsyncStreamAdvance(words * len(run)) if len(run) >= 8 else None
        rng_start = np.random.get_state()
        own_slots = [run[k]._slot for k in mine]
        if mine:
            num = den = np.zeros(len(mine), dtype=np.complex128)
            lam = np.zeros(len(mine), dtype=np.complex128)
            if is_eig:
                num, den = self.ctx.matvec_rayleigh(own_slots)
                for q in range(len(mine)):                              # AMS:264-268, the scalar arithmetic of the other path
                    lam[q] = complex(0.0, 0.0) if np.abs(den[q]) < 1e-12 else num[q] / den[q]
            stuck = np.array([run[k].stuck_counter for k in mine])
            psi0 = np.array([(base_psi * (10 ** (0 / 2.0)) * (10 ** (sc / 3.0))).real for sc in stuck])   # AMS:44
            pd = (rng_start, words, 0, np.asarray(mine, dtype=np.int32)) if pert == PERT_MT19937 else None
            self.ctx.pop_copy(POP_U, POP_X, own_slots)                  # snapshot (POP_U is unused by eig / linear problems)
            st = self.ctx.shifted_lu_solve(own_slots, lam, psi0, rhs_mode=0 if is_eig else 1, pert_mode=pert, pert_data=pd)
            alpha = np.array([complex(run[k].alpha_local_step) for k in mine], dtype=np.complex128)
            nrm = self.ctx.relax_normalise(own_slots, alpha, normalise=is_eig)
            res, fin = self.ctx.residual(KIND_EIG if is_eig else KIND_LINEAR, own_slots, lam if is_eig else None)
            loc = np.column_stack([num.real, num.imag, den.real, den.imag, st.astype(np.float64), nrm, res,
                                   np.asarray(fin, dtype=np.float64)])
        f = self._exchange(run, loc)
        status, nrm = f[:, 4], f[:, 5]
        bad = status != 0
        if is_eig:
            den = f[:, 2] + 1j * f[:, 3]
            bad = bad | (np.sqrt(den.real) < 1e-10) | ~(nrm > 1e-10)    # E2 (AMS:259), E4 (AMS:283)
        if bad.any():
            if ahead is not None:
                ahead.discard()
            if mine:
                self.ctx.pop_copy(POP_X, POP_U, own_slots)
            return False
        if is_eig:
            num = f[:, 0] + 1j * f[:, 1]
            for k, c in enumerate(run):
                c.lambda_k = complex(0.0, 0.0) if np.abs(den[k]) < 1e-12 else num[k] / den[k]
        for k, c in enumerate(run):
            c.local_psi_retries_needed = 0                              # attempts == 0 (AMS:278)
            c._invalidate()
            c.stuck_counter = max(0, c.stuck_counter - 1)               # AMS:286
            self._pre_resid[id(c)] = (f[k, 6], f[k, 7] != 0)
        if ahead is not None:
            ahead.apply()                                               # E3 of the whole run
        else:
            _advance_numpy_stream(words * len(run))
        return True

    def _gmres_batch(self, run, shift, psi0, stuck, is_eig, pert=PERT_NONE, pert_data=None):
        """First GMRES attempt of a run (AMS:60-90 with tol->rtol).  Returns ok[]."""
        ok = np.zeros(len(run), dtype=bool)
        cand_j = stuck > 1                                              # AMS:65
        dense = self.pert_matters(psi0) if pert != PERT_NONE else np.zeros(len(run), dtype=bool)
        L = np.nonzero(~dense)[0]
        if L.size:                                                      # random term below the rounding of a matvec
            use_j = np.zeros(L.size, dtype=np.int32)
            if np.any(cand_j[L]):
                okj = self.ctx.jacobi_check(shift[L], psi0[L])          # AMS:72
                use_j = (cand_j[L] & okj).astype(np.int32)
            info, inner, status = self.d_gmres([run[k] for k in L], shift[L], psi0[L], 0 if is_eig else 1, use_j)
            ok[L] = (info == 0) & (status == 0)
        D = np.nonzero(dense)[0]
        if D.size:                                                      # escalated psi: the reference's full H_solve
            if pert == PERT_MT19937:
                pd = (pert_data[0], pert_data[1], pert_data[2], np.asarray(pert_data[3])[D])
            else:
                pd = pert_data[D]
            info, inner, status = self.d_gmres_pert([run[k] for k in D], shift[D], psi0[D], 0 if is_eig else 1,
                                                    cand_j[D].astype(np.int32), pert, pd)
            ok[D] = (info == 0) & (status == 0)
        return ok

    def _attempt(self, c, method, shift, psi, is_eig, n, pert, stuck):
        """One solve attempt for one candidate on the device.  Raises like AMS:59/90/94-95."""
        pert_data = None
        if pert == PERT_UNIFORM:
            pert_data = np.empty((1, 2, n, n))
            pert_data[0, 0] = np.random.rand(n, n)                      # E3
            pert_data[0, 1] = np.random.rand(n, n)
        else:
            if pert == PERT_MT19937:
                pert_data = (np.random.get_state(), 4 * n * n, 0, np.zeros(1, dtype=np.int32))
            _advance_numpy_stream(4 * n * n)
        sh = np.array([shift], dtype=np.complex128)
        ps = np.array([psi.real if hasattr(psi, "real") else psi], dtype=np.float64)
        if method == DIRECT:
            st = self.d_lu_solve([c], sh, ps, 0 if is_eig else 1, pert, pert_data)[0]
            if st > 0:
                raise np.linalg.LinAlgError("Matrix is singular.")
            if st < 0:
                raise ValueError("array must not contain infs or NaNs" if st == -1 else "Solution vector not finite after solve.")
        elif method == GMRES:
            if self.gmres_compat == "scipy-legacy":
                raise TypeError("gmres() got an unexpected keyword argument 'tol'")
            if pert != PERT_NONE and self.pert_matters(ps)[0]:
                info, inner, status = self.d_gmres_pert([c], sh, ps, 0 if is_eig else 1,
                                                        np.array([1 if stuck > 1 else 0], dtype=np.int32), pert, pert_data)
            else:
                use_j = np.zeros(1, dtype=np.int32)
                if stuck > 1 and self.ctx.jacobi_check(sh, ps)[0]:
                    use_j[0] = 1
                info, inner, status = self.d_gmres([c], sh, ps, 0 if is_eig else 1, use_j)
            if status[0] == -1:
                raise ValueError("array must not contain infs or NaNs")
            if info[0] != 0:
                raise np.linalg.LinAlgError(f"GMRES did not converge cleanly (info={info[0]}).")
            if status[0] == -2:
                raise ValueError("Solution vector not finite after solve.")
        else:
            raise ValueError(f"Unknown solver method: {method}")

    def _ladder(self, c, shift, base_psi, max_attempts, pref, failed_method, is_eig, n, pert):
        """AMS:43-104 for one candidate whose batched attempt 0 (with `failed_method`) has already
        failed, followed by the success / failure branches of AMS:278-293."""
        from .solver import SolutionCandidate
        S = SolutionCandidate.State
        fallback = GMRES if pref == DIRECT else DIRECT
        attempts = 0
        method = failed_method
        # the failure that brought us here (AMS:99-103)
        if method == pref and pref != fallback and attempts == 0:
            method = fallback
            attempts = 0
        else:
            attempts += 1
        solved = False
        while attempts < max_attempts:
            psi = base_psi * (10 ** (attempts / 2.0)) * (10 ** (c.stuck_counter / 3.0))
            try:
                self._attempt(c, method, shift, psi, is_eig, n, pert, c.stuck_counter)
                solved = True
                break
            except (np.linalg.LinAlgError, ValueError, TypeError):
                if method == pref and pref != fallback and attempts == 0:
                    method = fallback
                    attempts = 0
                    continue
                attempts += 1
        if solved:
            c.local_psi_retries_needed = attempts
            alpha = np.array([complex(c.alpha_local_step)], dtype=np.complex128)
            nrm = self.d_relax([c], alpha, is_eig)
            c._invalidate()
            if is_eig and not (nrm[0] > 1e-10):
                c.v_k = (np.random.rand(n) + 1j * np.random.rand(n)) / np.sqrt(n)
                c._push(force=True)
            c.stuck_counter = max(0, c.stuck_counter - 1)
        else:                                                            # AMS:287-293
            c.stuck_counter += 1
            c.w_k *= 0.001
            c.alpha_local_step = max(c.alpha_local_step * 0.5, 1e-6)
            if c.stuck_counter >= MAX_STUCK_FOR_RETIREMENT:
                c.state = S.RETIRED
                c.num_resets += 1
            else:
                c.state = S.STUCK
                c.initialize_random_solution()                           # E5

    # ---- residual, histories, alpha/state adaptation, convergence (AMS:295-331) -------------
    def _finish(self, cands, A, b, strat, slots=None):
        """Residuals (AMS:295-301), histories (AMS:303-304), alpha / state adaptation (AMS:306-316) and the convergence test
        (AMS:318-331) of every stepped candidate, on the arrays of the candidate store: no Python object is touched per
        candidate (at 6144 candidates -- BASELINE configs[4] -- per-candidate NumPy scalar arithmetic was 40 ms per loop body
        in round 2, the per-candidate application of array decisions 8 ms in round 3)."""
        from .solver import ProblemType, SolutionCandidate
        S = SolutionCandidate.State
        st = self.store
        kind = cands[0].problem_type
        n = len(cands)
        if slots is None or len(slots) != n:
            slots = self._slots(cands)
        resv = np.empty(n, dtype=np.float64)
        okv = np.empty(n, dtype=bool)
        # residual against the construction-time matrix of each candidate: one group -- the whole list, no index
        # bookkeeping -- unless somebody mixed candidates of different matrix objects
        tags = st.mat_tag[slots]
        if (tags == tags[0]).all():
            groups = {0: (cands, slice(None))}
        else:
            groups = {}
            for t in np.unique(tags).tolist():
                ks = np.nonzero(tags == t)[0]
                groups[t] = ([cands[k] for k in ks.tolist()], ks)
        pre = getattr(self, "_pre_resid", None) or {}
        if pre and all(id(c) in pre for c in cands):
            # the owners' residuals came with the step's record (_share_step): no second exchange
            lamfin = np.isfinite(st.lam[slots]) if kind == ProblemType.EIGENVALUE else np.ones(n, dtype=bool)
            for k, c in enumerate(cands):
                r, fin = pre[id(c)]
                resv[k] = r
                okv[k] = bool(fin) and bool(lamfin[k])
            groups = {}
        self._pre_resid = {}
        for _, (grp, ix) in groups.items():
            self.bind_matrix(grp[0].problem_matrix)
            gsl = slots[ix]
            if kind == ProblemType.EIGENVALUE:
                lam = st.lam[gsl]                                          # complex(c.lambda_k) of every candidate
                res, fin = self.d_residual(KIND_EIG, grp, lam, gsl)
                fin = np.asarray(fin, dtype=bool) & np.isfinite(lam)
            elif kind == ProblemType.SOLVE_LINEAR_SYSTEM:
                res, fin = self.d_residual(KIND_LINEAR, grp, None, gsl)
                fin = np.asarray(fin, dtype=bool)
            else:
                sig = st.sig[gsl].astype(np.complex128)                    # complex(c.sigma_k)
                res, fin = self.d_residual(KIND_SVD, grp, sig, gsl)
                fin = np.asarray(fin, dtype=bool) & np.isfinite(sig)
            resv[ix] = res
            okv[ix] = fin
        st.set_real("res", slots, resv, F_NP)                              # residual_k = the np.float64 norm (AMS:295-301)
        self.bind_matrix(A)
        thr = strat.get("current_convergence_threshold", CONVERGENCE_RESIDUAL_TOL)
        self._log_history(cands, slots)                                    # AMS:303-304
        prev = st.prev[slots]
        # alpha is np.complex128(0.01) in the reference (AMS:17, imaginary part always 0) and keeps that type through
        # `alpha * 1.1` etc. until a clamp hands back the Python-float bound (AMS:308, 311, 314: min / max return one of their
        # arguments; NumPy orders complex scalars lexicographically) or convergence sets 0.0 (AMS:331); from then on it is a
        # Python float.  The store keeps the value and which of the two types it has.
        alpha = st.alpha[slots]
        akind = st.alpha_kind[slots]
        with np.errstate(invalid="ignore", over="ignore"):
            live = prev > 1e-10                                                  # AMS:306
            m1 = live & (resv < prev * 0.9)                                      # AMS:307 -> REFINING
            m2 = live & ~m1 & (resv > prev * 1.5) & (prev > 1e-5)                # AMS:310 -> STUCK
            m3 = live & ~m1 & ~m2                                                # AMS:313 -> EXPLORING
            raw = np.where(m1, alpha * 1.1, np.where(m2, alpha * 0.5, alpha * 0.95))
            clamped = np.where(m1, raw > 1.0, raw < 1e-6)                        # min(x, 1.0) is 1.0 iff 1.0 < x; max(x, 1e-6) is 1e-6 iff x < 1e-6
            new_alpha = np.where(clamped, np.where(m1, 1.0, 1e-6), raw)
            conv = (resv < thr) & okv                                            # AMS:318-331
        odd = akind == EXACT                       # an alpha of some other type that a caller assigned: that candidate alone, as objects
        stc = st.state[slots]
        CONV, STUCK, RETIRED = S.CONVERGED.value, S.STUCK.value, S.RETIRED.value
        open_ = stc != CONV
        new_state = stc.copy()
        new_state[m1 & open_] = S.REFINING.value
        new_state[m2 & open_] = STUCK
        new_state[m3 & open_ & (stc != STUCK) & (stc != RETIRED)] = S.EXPLORING.value
        new_state[conv] = CONV
        st.state[slots] = new_state
        upd = live & ~odd
        if upd.any():
            us = slots[upd]
            st.alpha[us] = new_alpha[upd]
            st.alpha_kind[us] = np.where(clamped[upd] | (akind[upd] == C_PY), C_PY, C_NP)
            st.alpha_obj[us] = MISSING
        for k in np.nonzero(odd & live)[0].tolist():
            c, a = cands[k], cands[k].alpha_local_step
            c.alpha_local_step = min(a * 1.1, 1.0) if m1[k] else (max(a * 0.5, 1e-6) if m2[k] else max(a * 0.95, 1e-6))
        if conv.any():
            cs = slots[conv]
            st.set_real("w", cs, 1.0, F_PY)
            st.stuck[cs] = 0
            st.alpha[cs] = 0.0
            st.alpha_kind[cs] = C_PY
            st.alpha_obj[cs] = MISSING
