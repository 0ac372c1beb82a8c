"""NumPy/SciPy test double of adaptive_matrix_solver_amd._cabi.Context -- TEST INFRASTRUCTURE.

Implements the phase API of the device context with the same NumPy/SciPy calls the reference
makes, so the product's HOST logic (engine.py / solver.py: RNG-event ordering, retry ladder,
bookkeeping, sharding) can be exercised on a machine without a GPU: `-m "not gpu"` tests
check it bit-for-bit against the fixtures captured from the reference, and the world_size-2
gloo tests use it to cover the N>1 path.  Never imported by the product package."""
import numpy as np
import scipy.linalg as sla

from oracle import maus_oracle as orc


class FakeContext:
    def __init__(self):
        self.rows = self.cols = 0
        self.A = None
        self.b = None
        self.V = None
        self.cap = 0
        self.pop = {}
        self.calls = {"lu": 0, "gemm": 0}

    # ---- data ------------------------------------------------------------------------------
    def set_matrix(self, A):
        A = np.ascontiguousarray(A, dtype=np.complex128)
        if A.shape != (self.rows, self.cols):
            self.pop = {}
            self.cap = 0
        self.A = A
        self.rows, self.cols = A.shape

    def set_rhs(self, b):
        self.b = np.ascontiguousarray(b, dtype=np.complex128)

    def set_eigvecs(self, V):
        self.V = np.ascontiguousarray(V, dtype=np.complex128)

    def pop_reserve(self, cap):
        if cap > self.cap:
            ld = max(self.rows, self.cols)
            for w in range(4):
                new = np.zeros((cap, ld), dtype=np.complex128)
                if w in self.pop:
                    new[: self.cap] = self.pop[w]
                self.pop[w] = new
            self.cap = cap

    def lu_reserve(self, n, count):
        return int(count)

    def pop_capacity(self):
        return self.cap

    def pop_put(self, which, slots, vecs):
        v = np.asarray(vecs, dtype=np.complex128)
        if v.ndim == 1:
            v = v[None]
        for k, s in enumerate(slots):
            self.pop[which][s, : v.shape[1]] = v[k]

    def pop_copy(self, which_dst, which_src, slots):
        for s in slots:
            self.pop[which_dst][s] = self.pop[which_src][s]

    def pop_get(self, which, slots, length):
        return np.array([self.pop[which][s, :length] for s in slots], dtype=np.complex128).reshape(len(slots), length)

    def device_info(self):
        return {"name": "fake (NumPy test double)", "cus": 0, "hbm_total": 0, "hbm_free": 0}

    def sync(self):
        pass

    def profile_enable(self, on=True):
        pass

    def profile_read(self):
        return {}

    # ---- phases (same arithmetic as the oracle / reference) -------------------------------------
    def matvec_rayleigh(self, slots):
        n = self.rows
        num = np.empty(len(slots), dtype=np.complex128)
        den = np.empty(len(slots), dtype=np.complex128)
        for k, s in enumerate(slots):
            v = self.pop[0][s, :n]
            den[k] = np.vdot(v, v)
            num[k] = np.vdot(v, self.A @ v)
        return num, den

    def shifted_lu_solve(self, slots, shift, psi, rhs_mode=0, pert_mode=0, pert_data=None):
        n = self.rows
        status = np.zeros(len(slots), dtype=np.int32)
        for k, s in enumerate(slots):
            self.calls["lu"] += 1
            target = self.A - shift[k] * np.eye(n, dtype=np.complex128) if rhs_mode == 0 else self.A
            ps = np.complex128(psi[k])
            if pert_mode == 1:
                pert = (pert_data[k, 0] - 0.5 + 1j * (pert_data[k, 1] - 0.5)) * ps * 0.15
                reg = ps * np.eye(n, dtype=np.complex128) + pert
            else:
                reg = ps * np.eye(n, dtype=np.complex128)
            H = target + reg
            rhs = self.pop[0][s, :n].copy() if rhs_mode == 0 else self.b
            try:
                x = sla.solve(H, rhs, assume_a="general")
                if not np.all(np.isfinite(x)):
                    status[k] = -2
                else:
                    self.pop[2][s, :n] = x
            except np.linalg.LinAlgError:
                status[k] = 1
            except ValueError:
                status[k] = -1
        return status

    def relax_normalise(self, slots, alpha, normalise=True):
        n = self.rows
        nrm = np.empty(len(slots))
        for k, s in enumerate(slots):
            a = np.complex128(alpha[k])
            v = (1.0 - a) * self.pop[0][s, :n] + a * self.pop[2][s, :n]
            nrm[k] = np.linalg.norm(v)
            if normalise and nrm[k] > 1e-10:
                v = v / nrm[k]
            self.pop[0][s, :n] = v
        return nrm

    def residual(self, kind, slots, lam=None):
        res = np.empty(len(slots))
        fin = np.empty(len(slots), dtype=bool)
        for k, s in enumerate(slots):
            if kind == 1:
                v = self.pop[0][s, : self.rows]
                res[k] = np.linalg.norm(self.A @ v - lam[k] * v)
                fin[k] = np.all(np.isfinite(v))
            elif kind == 2:
                x = self.pop[0][s, : self.rows]
                res[k] = np.linalg.norm(self.A @ x - self.b)
                fin[k] = np.all(np.isfinite(x))
            else:
                v = self.pop[0][s, : self.cols]
                u = self.pop[1][s, : self.rows]
                sg = lam[k].real
                res[k] = np.linalg.norm(self.A @ v - sg * u) + np.linalg.norm(self.A.conj().T @ u - sg * v)
                fin[k] = np.all(np.isfinite(v)) and np.all(np.isfinite(u))
        return res, fin

    def svd_power_propose(self, slots):
        out = np.empty((len(slots), 4))
        for k, s in enumerate(slots):
            v = self.pop[0][s, : self.cols]
            out[k, 0] = np.linalg.norm(v)
            t = self.A @ v
            s1 = np.linalg.norm(t)
            u = t / (s1 if s1 > 1e-10 else 1.0)
            out[k, 1] = s1
            out[k, 2] = np.linalg.norm(u)
            w = self.A.conj().T @ u
            s2 = np.linalg.norm(w)
            out[k, 3] = s2
            self.pop[3][s, : self.rows] = u                                   # proposal: u in POP_Y, v in POP_W
            self.pop[2][s, : self.cols] = w / (s2 if s2 > 1e-10 else 1.0)
        return out

    def svd_commit(self, slots):
        for s in slots:
            self.pop[1][s, : self.rows] = self.pop[3][s, : self.rows]
            self.pop[0][s, : self.cols] = self.pop[2][s, : self.cols]

    def svd_power_step(self, slots):
        out = self.svd_power_propose(slots)
        self.svd_commit(slots)
        return out

    def gram(self, which, slots, length):
        R = self.pop[which][list(slots), :length]
        return R.conj() @ R.T

    def herm_match(self, slots):
        n = self.rows
        idx = np.empty(len(slots), dtype=np.int32)
        nrm = np.empty(len(slots))
        for k, s in enumerate(slots):
            v = self.pop[0][s, :n]
            j = int(np.argmax(np.abs(v.conj().T @ self.V)))
            col = self.V[:, j].copy()
            nrm[k] = np.linalg.norm(col)
            self.pop[0][s, :n] = col / nrm[k]
            idx[k] = j
        return idx, nrm

    def jacobi_check(self, shift, psi):
        ok = np.zeros(len(shift), dtype=bool)
        d0 = np.diag(self.A)
        for k in range(len(shift)):
            d = (d0 - shift[k]) + psi[k]
            with np.errstate(all="ignore"):
                inv = 1.0 / d
            ok[k] = bool(np.all(np.isfinite(inv)) and np.all(np.abs(d) > 1e-12))
        return ok

    def gmres(self, slots, shift, psi, rhs_mode, use_jacobi, rtol=1e-8, restart=20, maxiter=50):
        n = self.rows
        info = np.zeros(len(slots), dtype=np.int32)
        inner = np.zeros(len(slots), dtype=np.int32)
        status = np.zeros(len(slots), dtype=np.int32)
        for k, s in enumerate(slots):
            H = (self.A - shift[k] * np.eye(n)) + psi[k] * np.eye(n)
            rhs = self.pop[0][s, :n].copy() if rhs_mode == 0 else self.b
            if not (np.all(np.isfinite(H)) and np.all(np.isfinite(rhs))):
                status[k] = -1
                continue
            inv_d = (1.0 / np.diag(H)) if use_jacobi[k] else None
            x, inf, inn, cyc = orc.gmres_restated(H, rhs, rhs, inv_d, rtol=rtol, maxiter=maxiter, restart=restart)
            info[k], inner[k] = inf, inn
            if inf == 0 and not np.all(np.isfinite(x)):
                status[k] = -2
            self.pop[2][s, :n] = x
        return info, inner, status

    def gmres_pert(self, slots, shift, psi, rhs_mode, want_jacobi, pert_mode, pert_data, rtol=1e-8, restart=20, maxiter=50):
        """NumPy double of maus_gmres_pert: GMRES against H_solve including the random term (AMS:49-52, 89)."""
        n = self.rows
        info = np.zeros(len(slots), dtype=np.int32)
        inner = np.zeros(len(slots), dtype=np.int32)
        status = np.zeros(len(slots), dtype=np.int32)
        jac = np.zeros(len(slots), dtype=bool)
        for k, s in enumerate(slots):
            target = self.A - shift[k] * np.eye(n, dtype=np.complex128) if rhs_mode == 0 else self.A
            ps = np.complex128(psi[k])
            reg = ps * np.eye(n, dtype=np.complex128)
            if pert_mode == 1:
                reg = reg + (pert_data[k, 0] - 0.5 + 1j * (pert_data[k, 1] - 0.5)) * ps * 0.15
            H = target + reg
            rhs = self.pop[0][s, :n].copy() if rhs_mode == 0 else self.b
            if not (np.all(np.isfinite(H)) and np.all(np.isfinite(rhs))):
                status[k] = -1
                continue
            inv_d = None
            if want_jacobi[k]:
                d = np.diag(H)
                with np.errstate(all="ignore"):
                    inv = 1.0 / d
                if np.all(np.isfinite(inv)) and np.all(np.abs(d) > 1e-12):
                    inv_d, jac[k] = inv, True
            x, inf, inn, cyc = orc.gmres_restated(H, rhs, rhs, inv_d, rtol=rtol, maxiter=maxiter, restart=restart)
            info[k], inner[k] = inf, inn
            if inf == 0 and not np.all(np.isfinite(x)):
                status[k] = -2
            self.pop[2][s, :n] = x
        return info, inner, status, jac

    def hist_append(self, which, slots, length):
        if not hasattr(self, "_hist"):
            self._hist = []
        first = len(self._hist)
        for sl in slots:
            self._hist.append(self.pop[which][sl].copy())
        return first

    def hist_get(self, indices, length):
        return np.array([self._hist[i][:length] for i in indices], dtype=np.complex128).reshape(len(indices), length)

    def hist_clear(self):
        self._hist = []
