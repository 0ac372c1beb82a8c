"""Ad-hoc boundary check of the direct path at n = 8192 (the largest order the panel kernels own)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptive_matrix_solver_amd import Context
from adaptive_matrix_solver_amd._cabi import PERT_MT19937, PERT_UNIFORM

n, P = 8192, 2
c = Context(0)
rng = np.random.default_rng(1)
A = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) / np.sqrt(n)
V = (rng.standard_normal((P, n)) + 1j * rng.standard_normal((P, n))) / np.sqrt(n)
c.set_matrix(A); c.pop_reserve(P); c.pop_put(0, [0, 1], V)
lam = np.array([0.3 + 0.1j, -0.2 + 0.4j]); psi = np.full(P, 1e-3)
np.random.seed(7); np.random.rand(101)
st = np.random.get_state()
U = np.empty((P, 2, n, n))
for k in range(P):
    U[k, 0] = np.random.rand(n, n); U[k, 1] = np.random.rand(n, n)
t = time.time()
s1 = c.shifted_lu_solve([0, 1], lam, psi, 0, PERT_UNIFORM, U); W1 = c.pop_get(2, [0, 1], n)
s2 = c.shifted_lu_solve([0, 1], lam, psi, 0, PERT_MT19937, (st, 4 * n * n, 0, np.arange(P, dtype=np.int32))); W2 = c.pop_get(2, [0, 1], n)
print("status", s1, s2, "equal", np.array_equal(W1, W2), "time", round(time.time() - t, 2), flush=True)
for k in range(P):
    H = A - lam[k] * np.eye(n) + psi[k] * np.eye(n) + 0.15 * psi[k] * ((U[k, 0] - 0.5) + 1j * (U[k, 1] - 0.5))
    r = np.linalg.norm(H @ W1[k] - V[k]); bound = 1e-13 * np.linalg.norm(H, 1) * np.linalg.norm(W1[k])
    print("cand", k, "residual", r, "bound", bound, "ok", r <= bound, flush=True)
