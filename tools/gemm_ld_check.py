"""zgemm rate against the leading dimension (L2 channel spread of the strided A rows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptive_matrix_solver_amd import Context
c = Context(0)
m, n, k, b = 2048, 2048, 256, 136
for ld in (4128, 4136, 4104, 4112, 4120, 4144, 4160, 4224):
    ms = c.zgemm_bench(m, n, k, ld, b, iters=3)
    print(f"ld={ld} (row stride mod 2KB = {ld * 16 % 2048:4d} B): {8.0 * m * n * k * b / ms * 1e-9:.1f} TF", flush=True)
