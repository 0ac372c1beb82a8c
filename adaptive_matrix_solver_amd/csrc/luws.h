// LU workspace descriptor shared by the C-ABI layer (capi.hip) and the LU driver (lu.hip).
#pragma once
#include "common.h"

// Tile-major storage of the LU workspace (round 3).  H and U are npad x (npad + 32) matrices stored as tiles of LU_TW = 64
// columns; inside a tile the npad rows follow each other at 64 elements (1 KB) each:
//     element (r, c)  ->  ((c / 64) * npad + r) * 64 + c % 64.
// Why: every kernel of the block-column factorisation (base panel, small triangular solves, the K = 16..128 update levels)
// works on 16..256 columns of ALL rows.  Row-major with a 66 KB row stride that is 256 B..4 KB pieces 66 KB apart -- each in
// another DRAM page and, after a few dozen rows, another 2 MB translation; tile-major the same pieces lie 1 KB apart in one
// contiguous 4 MB tile.  Measured with the K = 16 / 32 / 64 update shapes (tools/r03_ld_probe.py, 600 matrices so that nothing
// stays in the Infinity Cache): 2.4 -> 5.0, 3.3 -> 5.3, 4.1 -> 5.2 TB/s.  A 16-column panel, a 32-row triangular block, a
// K-tile of the zgemm never straddle a tile (all offsets are multiples of 16), so inside a tile they are plain row-major
// blocks with a leading dimension of 64 and only the tile base moves.
constexpr int LU_TW = 64;
static inline int lu_ntiles(int npad) { return (npad + 32 + LU_TW - 1) / LU_TW; }
__host__ __device__ static inline long lu_tile_off(long nrows, int col) { return ((long)(col >> 6) * nrows << 6) + (col & 63); }
__host__ __device__ static inline long lu_tix(long nrows, long r, int c) { return lu_tile_off(nrows, c) + (r << 6); }

struct LuWs {
    // ldh = npad + 32: logical column count (column npad carries the rhs); strideH = elements per matrix (tile-major:
    // lu_ntiles(npad) * npad * 64).  The GMRES path, which needs H_k as one dense row-major operand, builds it row-major
    // with leading dimension ldh into the same buffers (tiled = 0 in the build calls).
    c128* H; long ldh; long strideH; int n; int npad; int G;
    // implicit pivoting: rows never move.  perm[g][i] = physical row of H that holds logical row i; the finished rows of
    // U (and the carried right-hand side) are written in LOGICAL order to the second array U (same ld / stride as H)
    c128* U; int* perm;
    const int* ident = nullptr;  // identity row list (npad entries), for products whose rows are logical rows of U
    void* mw_sync = nullptr;     // per-matrix rendezvous area of the multi-workgroup panel; null unless this LU is alone on the device
    // multi-workgroup panel: how long a rendezvous may wait (100 MHz ticks) before the matrix is reported as
    // info = INT_MIN (the caller then repeats the batch with one workgroup per matrix), and the test hook that makes one
    // workgroup skip an arrival (MAUS_PANEL_MW_FORCE_ABORT)
    unsigned long long mw_timeout = 200000000ull;
    int mw_force_abort = 0;
    int* ipiv; int* info; int* flags;
    hipStream_t st;
    void (*tick)(void* ud, int klass, int phase, double flops, double bytes); void* ud;
};
