"""Round 3 probe: how much of the small-K LU update levels' time is the row stride?  The LU workspace is row-major with
a 66 KB row stride, so an update of a 16..128-column block touches 256 B..2 KB pieces of rows that lie 66 KB apart.
zgemm_bench embeds A (M x K), B (K x N), C (M x N) in one row-major array per matrix with leading dimension ld:
ld = 4128 is the LU workspace's; a small ld packs the same three operands into (M + K) x ld contiguous elements -- what a
column-tiled workspace would give.  G matrices per launch; G = 600 keeps the packed operands (> 600 MB) out of the 256 MB
Infinity Cache between launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptive_matrix_solver_amd import Context
c = Context(0)
M = 2304
for G in (181, 600):
    for K in (16, 32, 64, 128):
        N = K
        for ld in (4128, max(64, N + K)):
            ms = c.zgemm_bench(M, N, K, ld, G, iters=5)
            by = 16.0 * (M * K + K * N + 2.0 * M * N) * G
            print(f"G={G} M={M} N={N} K={K} ld={ld:5d}: {ms * 1e3:8.1f} us  {8.0 * M * N * K * G / ms * 1e-9:6.1f} TF(8MNK)  {by / (ms * 1e-3) * 1e-12:6.2f} TB/s", flush=True)
