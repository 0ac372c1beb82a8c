"""ctypes binding of libmaus_hip.so (include/maus_hip.h).

The product path has no CPU fallback: if the HIP library cannot be loaded, or a
device context cannot be created, this module raises -- it never routes the
candidate step through NumPy/SciPy."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmaus_hip.so")

# every symbol include/maus_hip.h declares (tests/test_cabi_symbols.py checks the list against the header)
SYMBOLS = [
    "maus_ctx_create", "maus_ctx_destroy", "maus_last_error", "maus_device_info", "maus_abi_version",
    "maus_set_matrix", "maus_set_rhs", "maus_pop_reserve", "maus_pop_capacity", "maus_pop_put", "maus_pop_get", "maus_pop_copy", "maus_pop_device_ptr", "maus_hist_append", "maus_hist_get", "maus_hist_clear", "maus_hist_generation",
    "maus_matvec_rayleigh", "maus_shifted_lu_solve", "maus_lu_reserve", "maus_lu_workspace_allocs", "maus_set_shared_device", "maus_lu_mw_aborts", "maus_relax_normalise", "maus_residual",
    "maus_svd_power_step", "maus_svd_power_propose", "maus_svd_commit", "maus_set_eigvecs", "maus_herm_match", "maus_herm_tridiag", "maus_herm_release", "maus_herm_tridiag_eig", "maus_herm_tridiag_eigvals", "maus_herm_backtransform", "maus_get_eigvecs", "maus_gmres", "maus_gmres_pert", "maus_jacobi_check",
    "maus_profile_union_ms", "maus_gram", "maus_zgemm_host", "maus_lu_solve_host", "maus_timer_start", "maus_timer_stop",
    "maus_profile_enable", "maus_profile_read", "maus_sync", "maus_mt19937_jump",
    "maus_device_count", "maus_comm_unique_id", "maus_comm_init", "maus_comm_destroy", "maus_comm_info",
    "maus_comm_allgather_records", "maus_comm_allgather_rows", "maus_comm_bcast", "maus_comm_bcast_eigvecs", "maus_comm_set_matrix", "maus_comm_stats",
]

COMM_ID_BYTES = 128

POP_X, POP_U, POP_W, POP_Y = 0, 1, 2, 3
KIND_EIG, KIND_LINEAR, KIND_SVD = 1, 2, 3
PERT_NONE, PERT_UNIFORM, PERT_MT19937 = 0, 1, 2
KC_NAMES = ["zgemm", "lu_panel", "trsm", "laswp", "build_h", "backsolve", "vector",
            "zgemm_k128", "zgemm_k64", "zgemm_k32", "zgemm_k16"]



class MtDesc(C.Structure):
    """maus_mt_desc (include/maus_hip.h)."""
    _fields_ = [("key", C.c_uint32 * 624), ("pos", C.c_int32), ("reserved", C.c_int32),
                ("words_per_candidate", C.c_uint64), ("lead_words", C.c_uint64), ("ordinals", C.POINTER(C.c_int32))]


_lib = None


class MausHipError(RuntimeError):
    pass


def load_library():
    """Load libmaus_hip.so (built in-tree by __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MausHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the candidate step.")
    lib = C.CDLL(LIB_PATH)
    vp, ip, dp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)
    i32p = C.POINTER(C.c_int32)
    sig = {
        "maus_ctx_create": ([C.c_int, C.POINTER(vp)], C.c_int),
        "maus_ctx_destroy": ([vp], C.c_int),
        "maus_last_error": ([vp], C.c_char_p),
        "maus_device_info": ([vp, C.c_char_p, C.c_int, ip, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)], C.c_int),
        "maus_abi_version": ([], C.c_int),
        "maus_set_matrix": ([vp, vp, C.c_int, C.c_int], C.c_int),
        "maus_set_rhs": ([vp, vp, C.c_int], C.c_int),
        "maus_pop_reserve": ([vp, C.c_int], C.c_int),
        "maus_pop_capacity": ([vp], C.c_int),
        "maus_pop_put": ([vp, C.c_int, vp, C.c_int, vp, C.c_int], C.c_int),
        "maus_pop_get": ([vp, C.c_int, vp, C.c_int, vp, C.c_int], C.c_int),
        "maus_pop_copy": ([vp, C.c_int, C.c_int, vp, C.c_int], C.c_int),
        "maus_pop_device_ptr": ([vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_long), ip], C.c_int),
        "maus_hist_append": ([vp, C.c_int, vp, C.c_int, C.c_int, C.POINTER(C.c_int64)], C.c_int),
        "maus_hist_get": ([vp, vp, C.c_int, C.c_int, vp], C.c_int),
        "maus_hist_clear": ([vp], C.c_int),
        "maus_hist_generation": ([vp], C.c_int64),
        "maus_matvec_rayleigh": ([vp, vp, C.c_int, vp, vp], C.c_int),
        "maus_shifted_lu_solve": ([vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp], C.c_int),
        "maus_lu_reserve": ([vp, C.c_int, C.c_int, ip], C.c_int),
        "maus_lu_workspace_allocs": ([vp], C.c_int),
        "maus_set_shared_device": ([vp, C.c_int], C.c_int),
        "maus_lu_mw_aborts": ([vp], C.c_int),
        "maus_relax_normalise": ([vp, vp, C.c_int, vp, C.c_int, vp], C.c_int),
        "maus_residual": ([vp, C.c_int, vp, C.c_int, vp, vp, vp], C.c_int),
        "maus_svd_power_step": ([vp, vp, C.c_int, vp], C.c_int),
        "maus_svd_power_propose": ([vp, vp, C.c_int, vp], C.c_int),
        "maus_svd_commit": ([vp, vp, C.c_int], C.c_int),
        "maus_set_eigvecs": ([vp, vp, C.c_int], C.c_int),
        "maus_herm_match": ([vp, vp, C.c_int, vp, vp], C.c_int),
        "maus_herm_tridiag": ([vp, vp, vp], C.c_int),
        "maus_herm_release": ([vp], C.c_int),
        "maus_herm_backtransform": ([vp, vp, C.c_int], C.c_int),
        "maus_herm_tridiag_eig": ([vp, vp, vp, C.c_int, vp, vp], C.c_int),
        "maus_herm_tridiag_eigvals": ([vp, vp, vp, C.c_int, vp], C.c_int),
        "maus_get_eigvecs": ([vp, vp, C.c_int], C.c_int),
        "maus_gmres": ([vp, vp, C.c_int, vp, vp, C.c_int, vp, C.c_double, C.c_int, C.c_int, vp, vp, vp], C.c_int),
        "maus_gmres_pert": ([vp, vp, C.c_int, vp, vp, C.c_int, vp, C.c_int, vp, C.c_double, C.c_int, C.c_int, vp, vp, vp, vp], C.c_int),
        "maus_jacobi_check": ([vp, C.c_int, vp, vp, vp], C.c_int),
        "maus_profile_union_ms": ([vp, C.c_int, C.POINTER(C.c_double)], C.c_int),
        "maus_gram": ([vp, C.c_int, vp, C.c_int, C.c_int, vp], C.c_int),
        "maus_zgemm_host": ([vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int], C.c_int),
        "maus_lu_solve_host": ([vp, C.c_int, C.c_int, vp, vp, vp, vp, vp], C.c_int),
        "maus_timer_start": ([vp], C.c_int),
        "maus_timer_stop": ([vp, C.POINTER(C.c_float)], C.c_int),
        "maus_profile_enable": ([vp, C.c_int], C.c_int),
        "maus_profile_read": ([vp, C.c_int, ip, dp, dp, dp], C.c_int),
        "maus_sync": ([vp], C.c_int),
        "maus_mt19937_jump": ([vp, i32p, C.c_uint64], C.c_int),
        "maus_device_count": ([], C.c_int),
        "maus_comm_unique_id": ([C.c_char_p], C.c_int),
        "maus_comm_init": ([vp, C.c_int, C.c_int, C.c_char_p], C.c_int),
        "maus_comm_destroy": ([vp], C.c_int),
        "maus_comm_info": ([vp, ip, ip], C.c_int),
        "maus_comm_allgather_records": ([vp, vp, C.c_size_t, vp], C.c_int),
        "maus_comm_allgather_rows": ([vp, C.c_int, vp, vp, C.c_int], C.c_int),
        "maus_comm_bcast": ([vp, vp, C.c_size_t, C.c_int], C.c_int),
        "maus_comm_bcast_eigvecs": ([vp, C.c_int, C.c_int], C.c_int),
        "maus_comm_set_matrix": ([vp, vp, C.c_int, C.c_int, C.c_int], C.c_int),
        "maus_comm_stats": ([vp, C.POINTER(C.c_long), dp, dp, C.c_int], C.c_int),
    }
    for name, (args, res) in sig.items():
        fn = getattr(lib, name)          # AttributeError here == missing export: fail loudly
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    return lib


def _c128(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.complex128)
    if shape is not None:
        assert a.shape == tuple(shape), (a.shape, shape)
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class _ClosedLib:
    """Stands in for the library handle of a closed Context: any call raises instead of handing a NULL context to C."""

    def __getattr__(self, name):
        raise MausHipError(f"{name}: the context has been closed")


class Context:
    """One device context (one per GPU).  Thin, typed wrappers over the C ABI."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.maus_ctx_create(device, C.byref(h))
        if rc != 0 or not h:
            msg = self.lib.maus_last_error(None)
            raise MausHipError(f"maus_ctx_create(device={device}) failed: {msg.decode() if msg else rc}; "
                               "a MI355X (gfx950) device is required -- no CPU fallback exists")
        self.h = h
        self.device = device
        self.rows = self.cols = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.maus_ctx_destroy(self.h)
            self.h = None
            self.lib = _ClosedLib()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if not getattr(self, "h", None):
            raise MausHipError(f"{what}: the context has been closed")
        if rc != 0:
            msg = self.lib.maus_last_error(self.h)
            raise MausHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")

    # -- info / measurement ------------------------------------------------
    def device_info(self):
        name = C.create_string_buffer(256)
        cus = C.c_int()
        tot, fr = C.c_size_t(), C.c_size_t()
        self._ck(self.lib.maus_device_info(self.h, name, 256, C.byref(cus), C.byref(tot), C.byref(fr)), "maus_device_info")
        return {"name": name.value.decode(), "cus": cus.value, "hbm_total": tot.value, "hbm_free": fr.value}

    def sync(self):
        self._ck(self.lib.maus_sync(self.h), "maus_sync")

    def timer_start(self):
        self._ck(self.lib.maus_timer_start(self.h), "maus_timer_start")

    def timer_stop(self) -> float:
        ms = C.c_float()
        self._ck(self.lib.maus_timer_stop(self.h, C.byref(ms)), "maus_timer_stop")
        return float(ms.value)

    def profile_enable(self, on=True):
        self._ck(self.lib.maus_profile_enable(self.h, int(on)), "maus_profile_enable")

    def profile_read_class(self, k):
        n = C.c_int()
        ms, fl, by, un = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        self._ck(self.lib.maus_profile_read(self.h, k, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)), "maus_profile_read")
        self._ck(self.lib.maus_profile_union_ms(self.h, k, C.byref(un)), "maus_profile_union_ms")
        return {"launches": n.value, "ms": ms.value, "flops": fl.value, "bytes": by.value, "union_ms": un.value}

    def profile_read(self):
        return {name: self.profile_read_class(k) for k, name in enumerate(KC_NAMES)}

    def lu_workspace_allocations(self) -> int:
        return int(self.lib.maus_lu_workspace_allocs(self.h))

    def set_shared_device(self, shared=True):
        """Tell the library that other processes use this GPU too (ranks of a gloo rehearsal on one device): kernels that
        need every one of their workgroups resident at once (the multi-workgroup LU panel) are then never used."""
        self._ck(self.lib.maus_set_shared_device(self.h, 1 if shared else 0), "maus_set_shared_device")

    def lu_mw_aborts(self) -> int:
        """Batches that were repeated with one panel workgroup per matrix after a rendezvous time-out."""
        return int(self.lib.maus_lu_mw_aborts(self.h))

    # -- problem data --------------------------------------------------------
    def set_matrix(self, A):
        A = _c128(A)
        assert A.ndim == 2
        self.rows, self.cols = A.shape
        self._ck(self.lib.maus_set_matrix(self.h, _ptr(A), A.shape[0], A.shape[1]), "maus_set_matrix")

    def set_rhs(self, b):
        b = _c128(b)
        self._ck(self.lib.maus_set_rhs(self.h, _ptr(b), b.shape[0]), "maus_set_rhs")

    def set_eigvecs(self, V):
        V = _c128(V)
        self._ck(self.lib.maus_set_eigvecs(self.h, _ptr(V), V.shape[0]), "maus_set_eigvecs")

    def get_eigvecs(self):
        n = self.rows
        V = np.empty((n, n), dtype=np.complex128)
        self._ck(self.lib.maus_get_eigvecs(self.h, _ptr(V), n), "maus_get_eigvecs")
        return V

    # -- population ----------------------------------------------------------
    def pop_reserve(self, cap):
        self._ck(self.lib.maus_pop_reserve(self.h, int(cap)), "maus_pop_reserve")

    def pop_capacity(self):
        return self.lib.maus_pop_capacity(self.h)

    @staticmethod
    def _slots(slots):
        return np.ascontiguousarray(slots, dtype=np.int32)

    def pop_put(self, which, slots, vecs):
        s = self._slots(slots)
        v = _c128(vecs)
        if v.ndim == 1:
            v = v[None, :]
        assert v.shape[0] == s.shape[0]
        self._ck(self.lib.maus_pop_put(self.h, which, _ptr(s), s.shape[0], _ptr(v), v.shape[1]), "maus_pop_put")

    def pop_get(self, which, slots, length):
        s = self._slots(slots)
        out = np.empty((s.shape[0], length), dtype=np.complex128)
        self._ck(self.lib.maus_pop_get(self.h, which, _ptr(s), s.shape[0], _ptr(out), length), "maus_pop_get")
        return out

    def pop_copy(self, which_dst, which_src, slots):
        s = self._slots(slots)
        self._ck(self.lib.maus_pop_copy(self.h, int(which_dst), int(which_src), _ptr(s), s.shape[0]), "maus_pop_copy")

    def pop_device_ptr(self, which):
        """(device address, leading dimension in complex elements, capacity in rows) of a population array."""
        ptr, ld, cap = C.c_void_p(), C.c_long(), C.c_int()
        self._ck(self.lib.maus_pop_device_ptr(self.h, int(which), C.byref(ptr), C.byref(ld), C.byref(cap)), "maus_pop_device_ptr")
        return int(ptr.value), int(ld.value), int(cap.value)

    def hist_append(self, which, slots, length) -> int:
        """Append rows `slots` of population array `which` to the device history; returns the first row's index."""
        s = self._slots(slots)
        first = C.c_int64()
        self._ck(self.lib.maus_hist_append(self.h, int(which), _ptr(s), s.shape[0], int(length), C.byref(first)), "maus_hist_append")
        return int(first.value)

    def hist_get(self, indices, length):
        ix = np.ascontiguousarray(indices, dtype=np.int64)
        out = np.empty((ix.shape[0], length), dtype=np.complex128)
        self._ck(self.lib.maus_hist_get(self.h, _ptr(ix), ix.shape[0], int(length), _ptr(out)), "maus_hist_get")
        return out

    def hist_clear(self):
        self._ck(self.lib.maus_hist_clear(self.h), "maus_hist_clear")

    def hist_generation(self) -> int:
        if not getattr(self, "h", None):
            raise MausHipError("maus_hist_generation: the context has been closed")
        return int(self.lib.maus_hist_generation(self.h))

    # -- phases --------------------------------------------------------------
    def matvec_rayleigh(self, slots):
        s = self._slots(slots)
        num = np.empty(s.shape[0], dtype=np.complex128)
        den = np.empty(s.shape[0], dtype=np.complex128)
        self._ck(self.lib.maus_matvec_rayleigh(self.h, _ptr(s), s.shape[0], _ptr(num), _ptr(den)), "maus_matvec_rayleigh")
        return num, den

    def _pert_arg(self, k, pert_mode, pert_data):
        """(keep-alive objects, void*) for the pert_data argument of the solve entry points."""
        if pert_mode == PERT_UNIFORM:
            pd = np.ascontiguousarray(pert_data, dtype=np.float64)
            assert pd.shape == (k, 2, self.rows, self.rows), pd.shape
            return (pd,), _ptr(pd)
        if pert_mode == PERT_MT19937:
            # pert_data = (numpy_state, words_per_candidate, lead_words, ordinals)
            st, wpc, lead, ords = pert_data
            ords = np.ascontiguousarray(ords, dtype=np.int32)
            assert ords.shape == (k,)
            pd = MtDesc()
            C.memmove(pd.key, np.ascontiguousarray(st[1], dtype=np.uint32).ctypes.data, 624 * 4)
            pd.pos = int(st[2]); pd.words_per_candidate = int(wpc); pd.lead_words = int(lead)
            pd.ordinals = ords.ctypes.data_as(C.POINTER(C.c_int32))
            return (pd, ords), C.cast(C.pointer(pd), C.c_void_p)
        return (), None

    def shifted_lu_solve(self, slots, shift, psi, rhs_mode=0, pert_mode=PERT_NONE, pert_data=None):
        s = self._slots(slots)
        k = s.shape[0]
        sh = _c128(shift, (k,))
        ps = np.ascontiguousarray(psi, dtype=np.float64)
        assert ps.shape == (k,)
        status = np.zeros(k, dtype=np.int32)
        keep, pdp = self._pert_arg(k, pert_mode, pert_data)
        self._ck(self.lib.maus_shifted_lu_solve(self.h, _ptr(s), k, _ptr(sh), _ptr(ps), int(rhs_mode), int(pert_mode),
                                                pdp, _ptr(status)), "maus_shifted_lu_solve")
        return status

    def lu_reserve(self, n, count) -> int:
        """Size the LU workspace once for `count` simultaneous n x n solves; returns its capacity in matrices."""
        cap = C.c_int()
        self._ck(self.lib.maus_lu_reserve(self.h, int(n), int(count), C.byref(cap)), "maus_lu_reserve")
        return int(cap.value)

    def relax_normalise(self, slots, alpha, normalise=True):
        s = self._slots(slots)
        al = _c128(alpha, (s.shape[0],))
        nrm = np.empty(s.shape[0], dtype=np.float64)
        self._ck(self.lib.maus_relax_normalise(self.h, _ptr(s), s.shape[0], _ptr(al), 1 if normalise else 0, _ptr(nrm)),
                 "maus_relax_normalise")
        return nrm

    def residual(self, kind, slots, lam=None):
        s = self._slots(slots)
        l = None if lam is None else _c128(lam, (s.shape[0],))
        res = np.empty(s.shape[0], dtype=np.float64)
        fin = np.empty(s.shape[0], dtype=np.int32)
        self._ck(self.lib.maus_residual(self.h, int(kind), _ptr(s), s.shape[0], _ptr(l), _ptr(res), _ptr(fin)), "maus_residual")
        return res, fin.astype(bool)

    def svd_power_step(self, slots):
        s = self._slots(slots)
        norms = np.empty((s.shape[0], 4), dtype=np.float64)
        self._ck(self.lib.maus_svd_power_step(self.h, _ptr(s), s.shape[0], _ptr(norms)), "maus_svd_power_step")
        return norms

    def herm_tridiag(self):
        """A = Q T Q^H of the bound Hermitian matrix on the device (zhetrd semantics): (d[n], e[n-1]) of the real T."""
        n = self.rows
        d = np.empty(n, dtype=np.float64)
        e = np.empty(max(n - 1, 1), dtype=np.float64)
        self._ck(self.lib.maus_herm_tridiag(self.h, _ptr(d), _ptr(e)), "maus_herm_tridiag")
        return d, e[: n - 1]

    @staticmethod
    def _scaled_tridiagonal(d, e):
        d = np.ascontiguousarray(d, dtype=np.float64)
        e = np.ascontiguousarray(e, dtype=np.float64)
        n = d.shape[0]
        if e.shape[0] != max(n - 1, 0):
            raise ValueError("tridiagonal matrix: e must have n - 1 entries")
        t = max(float(np.max(np.abs(d))) if n else 0.0, float(np.max(np.abs(e))) if n > 1 else 0.0)
        if not np.isfinite(t):
            raise ValueError("tridiagonal matrix is not finite")
        s = 2.0 ** int(np.floor(np.log2(t))) if t > 0.0 else 1.0           # a power of two: the scaling is exact
        return n, d / s, np.ascontiguousarray(e / s if n > 1 else np.zeros(1)), s

    def herm_release(self):
        """Hand back the reflector store of herm_tridiag (n x n) when no back-transformation will follow."""
        self._ck(self.lib.maus_herm_release(self.h), "maus_herm_release")

    def herm_tridiag_eigvals(self, d, e):
        """Eigenvalues (ascending) of the real symmetric tridiagonal T = (d, e) by bisection on the device."""
        n, ds, es, s = self._scaled_tridiagonal(d, e)
        w = np.empty(n, dtype=np.float64)
        self._ck(self.lib.maus_herm_tridiag_eigvals(self.h, _ptr(ds), _ptr(es), n, _ptr(w)), "maus_herm_tridiag_eigvals")
        return w * s

    def herm_tridiag_eig(self, d, e):
        """Eigenvalues (ascending) of the real symmetric tridiagonal T = (d, e) by bisection on the device, its eigenvectors
        (twisted factorisation, no reorthogonalisation) left there for herm_backtransform(None).  Returns (w, diag) with diag =
        (smallest gap / ||T||, largest residual component / ||T||, ||T||): see engine.device_eigh for the acceptance rule.
        T is scaled by a power of two to ||T|| ~ 1 first (exact), so the squares of the off-diagonal stay in range."""
        n, ds, es, s = self._scaled_tridiagonal(d, e)
        w = np.empty(n, dtype=np.float64)
        diag = np.empty(3, dtype=np.float64)
        self._ck(self.lib.maus_herm_tridiag_eig(self.h, _ptr(ds), _ptr(es), n, _ptr(w), _ptr(diag)), "maus_herm_tridiag_eig")
        return w * s, (float(diag[0]), float(diag[1]), float(diag[2]) * s)

    def herm_backtransform(self, Z):
        """V = Q Z on the device from the real eigenvectors Z[n][n] of T (None: those herm_tridiag_eig left on the device); V
        becomes the context's eigenvector matrix.  A column-major Z (what LAPACK returns) is passed as it is and transposed on
        the device."""
        if Z is None:
            self._ck(self.lib.maus_herm_backtransform(self.h, None, 0), "maus_herm_backtransform")
            return
        Z = np.asarray(Z, dtype=np.float64)
        if Z.shape != (self.rows, self.rows):
            raise ValueError("herm_backtransform: Z must be n x n")
        if Z.flags.f_contiguous and not Z.flags.c_contiguous:
            self._ck(self.lib.maus_herm_backtransform(self.h, _ptr(Z.T), 1), "maus_herm_backtransform")
        else:
            Z = np.ascontiguousarray(Z)
            self._ck(self.lib.maus_herm_backtransform(self.h, _ptr(Z), 0), "maus_herm_backtransform")

    def svd_power_propose(self, slots):
        """The power step without its effect: norms as svd_power_step, proposed u / v left in POP_Y / POP_W."""
        s = self._slots(slots)
        norms = np.empty((s.shape[0], 4), dtype=np.float64)
        self._ck(self.lib.maus_svd_power_propose(self.h, _ptr(s), s.shape[0], _ptr(norms)), "maus_svd_power_propose")
        return norms

    def svd_commit(self, slots):
        s = self._slots(slots)
        if s.shape[0]:
            self._ck(self.lib.maus_svd_commit(self.h, _ptr(s), s.shape[0]), "maus_svd_commit")

    def herm_match(self, slots):
        s = self._slots(slots)
        idx = np.empty(s.shape[0], dtype=np.int32)
        nrm = np.empty(s.shape[0], dtype=np.float64)
        self._ck(self.lib.maus_herm_match(self.h, _ptr(s), s.shape[0], _ptr(idx), _ptr(nrm)), "maus_herm_match")
        return idx, nrm

    def gram(self, which, slots, length):
        """G[i, j] = np.vdot(x_i, x_j) for the rows `slots` of population array `which` (first `length` entries)."""
        s = self._slots(slots)
        k = s.shape[0]
        out = np.empty((k, k), dtype=np.complex128)
        self._ck(self.lib.maus_gram(self.h, int(which), _ptr(s), k, int(length), _ptr(out)), "maus_gram")
        return out

    def gmres(self, slots, shift, psi, rhs_mode, use_jacobi, rtol=1e-8, restart=20, maxiter=50):
        s = self._slots(slots)
        k = s.shape[0]
        sh = _c128(shift, (k,))
        ps = np.ascontiguousarray(psi, dtype=np.float64)
        uj = np.ascontiguousarray(use_jacobi, dtype=np.int32)
        info = np.zeros(k, dtype=np.int32)
        inner = np.zeros(k, dtype=np.int32)
        status = np.zeros(k, dtype=np.int32)
        self._ck(self.lib.maus_gmres(self.h, _ptr(s), k, _ptr(sh), _ptr(ps), int(rhs_mode), _ptr(uj), float(rtol),
                                     int(restart), int(maxiter), _ptr(info), _ptr(inner), _ptr(status)), "maus_gmres")
        return info, inner, status

    def gmres_pert(self, slots, shift, psi, rhs_mode, want_jacobi, pert_mode, pert_data, rtol=1e-8, restart=20, maxiter=50):
        """GMRES against the materialised H_k including the random term of AMS:49-50 -> (info, inner, status, jacobi_used)."""
        s = self._slots(slots)
        k = s.shape[0]
        sh = _c128(shift, (k,))
        ps = np.ascontiguousarray(psi, dtype=np.float64)
        wj = np.ascontiguousarray(want_jacobi, dtype=np.int32)
        info = np.zeros(k, dtype=np.int32)
        inner = np.zeros(k, dtype=np.int32)
        status = np.zeros(k, dtype=np.int32)
        jac = np.zeros(k, dtype=np.int32)
        keep, pdp = self._pert_arg(k, pert_mode, pert_data)
        self._ck(self.lib.maus_gmres_pert(self.h, _ptr(s), k, _ptr(sh), _ptr(ps), int(rhs_mode), _ptr(wj), int(pert_mode), pdp,
                                          float(rtol), int(restart), int(maxiter), _ptr(info), _ptr(inner), _ptr(status),
                                          _ptr(jac)), "maus_gmres_pert")
        return info, inner, status, jac.astype(bool)

    def jacobi_check(self, shift, psi):
        sh = _c128(shift)
        k = sh.shape[0]
        ps = np.ascontiguousarray(psi, dtype=np.float64)
        ok = np.zeros(k, dtype=np.int32)
        self._ck(self.lib.maus_jacobi_check(self.h, k, _ptr(sh), _ptr(ps), _ptr(ok)), "maus_jacobi_check")
        return ok.astype(bool)

    # -- population sharding: RCCL collectives on this context's stream (csrc/comm.hip) -----------------------------
    def comm_init(self, rank: int, world: int, unique_id: bytes):
        assert len(unique_id) == COMM_ID_BYTES
        self._ck(self.lib.maus_comm_init(self.h, int(rank), int(world), unique_id), "maus_comm_init")

    def comm_destroy(self):
        self._ck(self.lib.maus_comm_destroy(self.h), "maus_comm_destroy")

    def comm_allgather_records(self, send: np.ndarray, world: int) -> np.ndarray:
        """send: C-contiguous array (any dtype) of the same shape on every rank -> array of shape (world,) + send.shape."""
        send = np.ascontiguousarray(send)
        out = np.empty((world,) + send.shape, dtype=send.dtype)
        self._ck(self.lib.maus_comm_allgather_records(self.h, _ptr(send), send.nbytes, _ptr(out)), "maus_comm_allgather_records")
        return out

    def comm_allgather_rows(self, which, slots_by_rank, length):
        counts = np.ascontiguousarray([len(s) for s in slots_by_rank], dtype=np.int32)
        flat = np.ascontiguousarray([s for sl in slots_by_rank for s in sl], dtype=np.int32)
        self._ck(self.lib.maus_comm_allgather_rows(self.h, int(which), _ptr(flat), _ptr(counts), int(length)), "maus_comm_allgather_rows")

    def comm_bcast(self, arr: np.ndarray, root=0):
        """In-place broadcast of a C-contiguous array (same shape / dtype on every rank)."""
        assert arr.flags["C_CONTIGUOUS"]
        self._ck(self.lib.maus_comm_bcast(self.h, _ptr(arr), arr.nbytes, int(root)), "maus_comm_bcast")
        return arr

    def comm_set_matrix(self, A, rank, root=0, resident_on_root=False):
        """set_matrix of a sharded run: `root` uploads A (or, resident_on_root, broadcasts the copy its device already
        holds), the other ranks receive it device to device (only the shape of their A is read)."""
        upload = rank == root and not resident_on_root
        if upload:
            A = _c128(A)
        self._ck(self.lib.maus_comm_set_matrix(self.h, _ptr(A) if upload else None, A.shape[0], A.shape[1], int(root)),
                 "maus_comm_set_matrix")
        self.rows, self.cols = A.shape

    def comm_bcast_eigvecs(self, n, root=0):
        self._ck(self.lib.maus_comm_bcast_eigvecs(self.h, int(n), int(root)), "maus_comm_bcast_eigvecs")

    def comm_stats(self, reset=False):
        calls, by, ms = C.c_long(), C.c_double(), C.c_double()
        self._ck(self.lib.maus_comm_stats(self.h, C.byref(calls), C.byref(by), C.byref(ms), 1 if reset else 0), "maus_comm_stats")
        return {"collectives": int(calls.value), "bytes": float(by.value), "ms": float(ms.value)}

    # -- test / utility entry points ------------------------------------------
    def zgemm(self, A, B, C_in=None, b_layout=0, conj_a=False, conj_b=False, alpha=1.0, beta=0):
        A = _c128(A)
        B = _c128(B)
        M, K = A.shape
        N = B.shape[0] if b_layout else B.shape[1]
        Cm = np.zeros((M, N), dtype=np.complex128) if C_in is None else _c128(C_in).copy()
        self._ck(self.lib.maus_zgemm_host(self.h, M, N, K, _ptr(A), _ptr(B), _ptr(Cm), int(b_layout), int(conj_a),
                                          int(conj_b), float(alpha), int(beta)), "maus_zgemm_host")
        return Cm

    def lu_solve(self, A, b, want_ipiv=False):
        A = _c128(A)
        b = _c128(b)
        if A.ndim == 2:
            A = A[None]
            b = b[None]
        cnt, n, _ = A.shape
        x = np.empty((cnt, n), dtype=np.complex128)
        status = np.zeros(cnt, dtype=np.int32)
        ipiv = np.zeros((cnt, n), dtype=np.int32) if want_ipiv else None
        self._ck(self.lib.maus_lu_solve_host(self.h, cnt, n, _ptr(A), _ptr(b), _ptr(x), _ptr(status), _ptr(ipiv)), "maus_lu_solve_host")
        return (x, status, ipiv) if want_ipiv else (x, status)


def device_count() -> int:
    """HIP devices visible to this process (0 on a box without a GPU)."""
    return int(load_library().maus_device_count())


def comm_unique_id() -> bytes:
    """A fresh RCCL unique id (one rank creates it, every rank passes it to Context.comm_init)."""
    lib = load_library()
    buf = C.create_string_buffer(COMM_ID_BYTES)
    if lib.maus_comm_unique_id(buf) != 0:
        msg = lib.maus_last_error(None)
        raise MausHipError(f"maus_comm_unique_id failed: {msg.decode() if msg else ''}")
    return buf.raw


def mt19937_jump(key: np.ndarray, pos: int, nwords: int):
    """Advance a legacy NumPy MT19937 (key[624], pos) by nwords 32-bit outputs."""
    lib = load_library()
    k = np.ascontiguousarray(key, dtype=np.uint32).copy()
    p = C.c_int32(int(pos))
    rc = lib.maus_mt19937_jump(_ptr(k), C.byref(p), C.c_uint64(int(nwords)))
    if rc != 0:
        raise MausHipError("maus_mt19937_jump failed")
    return k, int(p.value)
