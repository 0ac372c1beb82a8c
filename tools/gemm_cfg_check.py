"""Time + check one zgemm configuration (MAUS_GEMM_CFG is read once per process)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptive_matrix_solver_amd import Context

c = Context(0)
rng = np.random.default_rng(5)
M, N, K = 200, 136, 96
A = rng.standard_normal((M, K)) + 1j * rng.standard_normal((M, K))
B = rng.standard_normal((K, N)) + 1j * rng.standard_normal((K, N))
C0 = rng.standard_normal((M, N)) + 1j * rng.standard_normal((M, N))
got = c.zgemm(A, B, C_in=C0, alpha=-1.0, beta=1)
ref = C0 - A @ B
err = np.abs(got - ref).max() / np.abs(ref).max()
out = [f"cfg={os.environ.get('MAUS_GEMM_CFG', '0')} relerr={err:.2e}"]
for (m, n, k, b) in [(3840, 3840, 256, 136), (2048, 2048, 256, 136), (1024, 1024, 256, 136), (2048, 2048, 128, 136)]:
    ms = c.zgemm_bench(m, n, k, 4128, b, iters=3)
    out.append(f"{m}x{n}x{k}:{8.0 * m * n * k * b / ms * 1e-9:.1f}TF")
print(" ".join(out), flush=True)
