"""zgemm rate on the trailing-update shapes of the 512-wide outer block (K = 512)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptive_matrix_solver_amd import Context
c = Context(0)
out = []
for (m, n, k, b) in [(3584, 3616, 512, 136), (2048, 2080, 512, 136), (1024, 1056, 512, 136), (3584, 3616, 512, 271)]:
    ms = c.zgemm_bench(m, n, k, 4128, b, iters=3)
    out.append(f"{m}x{n}x{k}x{b}:{8.0 * m * n * k * b / ms * 1e-9:.1f}TF")
print(os.environ.get("TAG", ""), " ".join(out), flush=True)
