"""Structure-of-arrays state of the candidate population (round 4).

The reference keeps the per-candidate bookkeeping of AMS:107-127 -- residual, previous residual, step size, state, weight,
stuck counter, lambda / sigma -- in attributes of one Python object per candidate and walks the objects in every phase of
the loop body (AMS:295-331, 424-475, 504-549).  At BASELINE configs[4] (6 144 candidates) those walks were 20 of the 33 ms
of a loop body.  Here the values live in arrays indexed by the candidate's device slot; `SolutionCandidate` exposes them
under the reference's attribute names (solver.py), and the engine and the solver read and write whole populations with
array operations.

Types are part of the reference's behaviour (`alpha_local_step` is np.complex128 until a clamp hands back a Python
float, `residual_k` starts as float('inf') and becomes np.float64, ...), so every value carries a KIND that says which
scalar type the attribute has, and anything of another type a caller assigns is kept as the object it was (kind EXACT,
with its numeric value mirrored for the array code).  `*_obj` caches the object last handed out or assigned."""
import numpy as np

HREF = "\x00maus-history-row"        # first item of a compact history reference (solver._LazyHistory)
MISSING = type("_Missing", (), {"__repr__": lambda self: "<not materialised>"})()

EXACT = 255
# kinds of the real-valued attributes (residual_k, prev_residual, w_k, sigma_k)
F_NP, F_PY = 0, 1               # np.float64 / Python float
# kinds of alpha_local_step (imaginary part always 0: AMS:17, 308-314) and lambda_k
C_NP, C_PY = 0, 1               # np.complex128 / for alpha: Python float, for lambda_k: Python complex

_FIELDS = (
    # name, dtype, default
    ("res", np.float64, np.inf), ("res_kind", np.uint8, F_PY),
    ("prev", np.float64, np.inf), ("prev_kind", np.uint8, F_PY),
    ("alpha", np.float64, 0.0), ("alpha_kind", np.uint8, C_NP),
    ("w", np.float64, 0.0), ("w_kind", np.uint8, F_PY),
    ("lam", np.complex128, np.nan), ("lam_kind", np.uint8, EXACT),
    ("sig", np.float64, np.nan), ("sig_kind", np.uint8, EXACT),
    ("state", np.uint8, 1),
    ("stuck", np.int64, 0), ("retries", np.int64, 0), ("resets", np.int64, 0),
    ("host_valid", np.bool_, True), ("dev_valid", np.bool_, False),
    ("records", np.bool_, True),            # param_history records every iterate (record_history=False: residual_history only)
    ("mat_tag", np.int32, 0),               # which matrix object the candidate was constructed on
)
_OBJ_FIELDS = ("res_obj", "prev_obj", "alpha_obj", "w_obj", "lam_obj", "sig_obj", "b_obj")


class CandidateStore:
    def __init__(self, cap: int = 256):
        self.cap = 0
        self.hist_log = []              # one HistoryRecord per batched step whose candidates keep their iterates on the device
        self.ensure(cap)

    def ensure(self, cap: int) -> None:
        if cap <= self.cap:
            return
        cap = max(cap, 2 * self.cap, 64)
        for name, dtype, fill in _FIELDS:
            new = np.full(cap, fill, dtype=dtype)
            if self.cap:
                new[: self.cap] = getattr(self, name)
            setattr(self, name, new)
        for name in _OBJ_FIELDS:
            new = np.empty(cap, dtype=object)
            new[:] = MISSING
            if self.cap:
                new[: self.cap] = getattr(self, name)
            setattr(self, name, new)
        self.cap = cap

    def init_slot(self, s: int) -> None:
        """The state of a freshly constructed candidate (AMS:113-126); lambda_k / sigma_k / w_k / alpha are assigned by
        SolutionCandidate.__init__ through the attribute setters."""
        self.ensure(s + 1)
        for name, _dtype, fill in _FIELDS:
            getattr(self, name)[s] = fill
        for name in _OBJ_FIELDS:
            getattr(self, name)[s] = MISSING
        self.b_obj[s] = None
        self.lam_obj[s] = None
        self.sig_obj[s] = None

    # ---- whole-population writes by the engine (slots: integer array) --------------------------------------
    def set_real(self, name: str, slots, values, kind: int) -> None:
        getattr(self, name)[slots] = values
        getattr(self, name + "_kind")[slots] = kind
        getattr(self, name + "_obj")[slots] = MISSING

    def copy_real(self, dst: str, src: str, slots) -> None:
        getattr(self, dst)[slots] = getattr(self, src)[slots]
        getattr(self, dst + "_kind")[slots] = getattr(self, src + "_kind")[slots]
        getattr(self, dst + "_obj")[slots] = getattr(self, src + "_obj")[slots]

    def set_all(self, name: str, slots, obj) -> None:
        """The same object for every slot of an object array (an ndarray must not be broadcast element-wise)."""
        box = np.empty(1, dtype=object)
        box[0] = obj
        getattr(self, name)[slots] = box


def real_value(num, kind, obj, k):
    """The attribute value of entry k of snapshotted (numeric, kind, object) arrays of a real-valued attribute."""
    kd = kind[k]
    if kd == F_NP:
        return num[k]
    if kd == F_PY:
        return float(num[k])
    return obj[k]


def complex_value(num, kind, obj, k):
    """Same for lambda_k: np.complex128 / Python complex / the assigned object."""
    kd = kind[k]
    if kd == C_NP:
        return num[k]
    if kd == C_PY:
        return complex(num[k])
    return obj[k]


class HistoryRecord:
    """One batched step's contribution to the histories of its candidates (AMS:303-304): residual_k of every stepped
    candidate and, for those that record iterates, the compact reference (HREF, scalar, generation, row, length[, row,
    length]) into the device history store.  Snapshots of the store's arrays; a candidate appends its own entry when its
    history is read (SolutionCandidate._replay_history)."""
    __slots__ = ("slots", "inv", "res", "scal", "scal_complex", "hidx", "href")

    def __init__(self, slots, res, scal, scal_complex, hidx, href):
        self.slots = slots
        self.inv = None
        self.res = res                  # (numeric, kind, object) snapshots
        self.scal = scal                # same for lambda_k / sigma_k, or None (linear systems record (x,))
        self.scal_complex = scal_complex
        self.hidx = hidx                # position among the recording candidates, -1 for the others
        self.href = href                # (generation, iu, lu, iv, lv) or (generation, iv, lv); None: nobody records

    def replay(self, slot, rh, ph):
        inv = self.inv
        if inv is None:
            inv = self.inv = np.full(int(self.slots.max()) + 1 if len(self.slots) else 0, -1, dtype=np.int64)
            inv[self.slots] = np.arange(len(self.slots))
        if slot >= len(inv):
            return
        k = inv[slot]
        if k < 0:
            return
        if ph is not None and self.href is not None:
            j = int(self.hidx[k])
            if j >= 0:
                if self.scal is None:
                    scalar = None
                elif self.scal_complex:
                    scalar = complex_value(*self.scal, k)
                else:
                    scalar = real_value(*self.scal, k)
                h = self.href
                if len(h) == 5:
                    ph.append((HREF, scalar, h[0], h[1] + j, h[2], h[3] + j, h[4]))
                else:
                    ph.append((HREF, scalar, h[0], h[1] + j, h[2]))
        rh.append(real_value(*self.res, k))
