"""GEMM-only sweep on the GPU: LU trailing-update shapes, device-resident (tools, not product)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptive_matrix_solver_amd import Context
c = Context(0)
shapes = [(2048, 2048, 256, 1), (2048, 2048, 256, 16), (2048, 2048, 256, 136), (3840, 3872, 256, 136), (1024, 1056, 256, 136),
          (2048, 2048, 128, 136), (2048, 2048, 64, 136), (2048, 2048, 32, 136), (4064, 32, 32, 136), (128, 3840, 128, 136)]
for M, N, K, b in shapes:
    ld = 4128
    ms = c.zgemm_bench(M, N, K, ld, b, iters=3)
    print(f"M={M:5d} N={N:5d} K={K:4d} batch={b:4d}: {ms:9.3f} ms  {8.0*M*N*K*b/ms*1e-9:7.2f} TFLOP/s", flush=True)
