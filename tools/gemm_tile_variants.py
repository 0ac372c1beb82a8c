"""zgemm rate (8MNK-equivalent TFLOP/s) of LU-update shapes for the tile variants selected by MAUS_GEMM_DMA."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptive_matrix_solver_amd import Context
c = Context(0)
out = []
for (m, n, k, b) in [(3584, 3616, 512, 60), (1024, 1056, 512, 60), (512, 544, 512, 60), (256, 3000, 256, 60), (3584, 256, 256, 60), (1024, 256, 256, 60), (2048, 128, 128, 60), (128, 3000, 128, 60)]:
    ms = c.zgemm_bench(m, n, k, 4128, b, iters=3)
    out.append(f"{m}x{n}x{k}:{8.0 * m * n * k * b / ms * 1e-9:.1f}")
print(f"DMA={os.environ.get('MAUS_GEMM_DMA', '1'):>3s}", " ".join(out), flush=True)
