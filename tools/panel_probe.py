import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from adaptive_matrix_solver_amd import Context
c = Context(0)
n, cnt = 4096, int(os.environ.get("CNT", "64"))
rng = np.random.default_rng(0)
A = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) / np.sqrt(n)
c.set_matrix(A)
c.pop_reserve(cnt)
V = rng.standard_normal((cnt, n)) + 1j * rng.standard_normal((cnt, n))
c.pop_put(0, list(range(cnt)), V)
shift = (rng.standard_normal(cnt) + 1j * rng.standard_normal(cnt)) * 0.3
c.shifted_lu_solve(list(range(cnt)), shift, np.full(cnt, 1e-20))
c.profile_enable(True)
c.shifted_lu_solve(list(range(cnt)), shift, np.full(cnt, 1e-20))
p = c.profile_read()
print("DBG", os.environ.get("MAUS_PANEL_DBG", "0"), "cnt", cnt, {k: (round(v["ms"], 1), round(v["flops"] / max(v["ms"], 1e-9) * 1e-9, 1)) for k, v in p.items()})
