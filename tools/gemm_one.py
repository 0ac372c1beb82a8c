import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptive_matrix_solver_amd import Context
c = Context(0)
M, N, K, b = 2048, 2048, 256, 136
ms = c.zgemm_bench(M, N, K, 4128, b, iters=2)
print(f"{ms:.3f} ms {8.0*M*N*K*b/ms*1e-9:.2f} TF", flush=True)
