import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptive_matrix_solver_amd import Context
c = Context(0)
for M, N, K, b in [(2048, 2048, 256, 16), (2048, 2048, 256, 136), (3840, 3872, 256, 136)]:
    ms = c.zgemm_bench(M, N, K, 4128, b, iters=5)
    print(f"M={M:5d} N={N:5d} K={K:4d} batch={b:4d}: {ms:9.3f} ms  {8.0*M*N*K*b/ms*1e-9:7.2f} TFLOP/s", flush=True)
