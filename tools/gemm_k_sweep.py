"""zgemm rate of the LU-update shape against K (how much of a tile's time is prologue / epilogue)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptive_matrix_solver_amd import Context
c = Context(0)
out = []
for (m, n, k, b) in [(2048, 2048, 128, 96), (2048, 2048, 256, 96), (2048, 2048, 512, 96), (2048, 2048, 1024, 96), (2048, 2048, 2048, 64)]:
    ms = c.zgemm_bench(m, n, k, 2048 + k + 64, b, iters=3)
    out.append(f"K={k}:{8.0 * m * n * k * b / ms * 1e-9:.1f}TF")
print(os.environ.get("TAG", ""), " ".join(out), flush=True)
