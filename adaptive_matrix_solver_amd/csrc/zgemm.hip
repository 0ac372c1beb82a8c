// Batched complex128 GEMM on v_mfma_f64_16x16x4_f64 (gfx950), the dense contraction of
// the MAUS hot path:
//   * LU trailing update  C -= L21 * U12            (AMS:59 -> zgetrf)
//   * population matvec   Y  = X * A^T  (= A@v per candidate row; AMS:268, 297)
//   * SVD                 S  = U * conj(A)          (A^H u per candidate; AMS:240, 301)
//   * Hermitian match     S  = conj(X) * V          (AMS:165)
//
// C[M,N] = alpha * opA(A)[M,K] * opB(B)[K,N] + beta * C     (alpha = +-1, beta in {0,1})
// A is row-major [m][k].  B is row-major [k][n] (BLAY=0) or [n][k] (BLAY=1, dot-product form).
//
// Complex product on real MFMAs (4M form, same rounding structure as a scalar FMA chain):
//   Cre += Are*Bre ; Cre += (-Aim)*Bim   (BLGP bit0 = negate A, measured on gfx950)
//   Cim += Are*Bim ; Cim += Aim*Bre
// Operand / result lane maps of v_mfma_f64_16x16x4_f64 (verified by tools/probe_mfma_f64):
//   a = A[row = lane&15][k = lane>>4], b = B[k = lane>>4][col = lane&15],
//   d[r] = D[row = (lane>>4) + 4r][col = lane&15].
//
// Tiling: BM x BN block tile, BK = 16, tiles staged through LDS as [k][m] / [k][n]
// (k-major, XOR-swizzled, see the kernel) so every fragment is one ds_read_b128 of an
// interleaved (re,im) pair; the next K-tile is prefetched into registers while the
// current one feeds the MFMAs.  The fp64 matrix pipe sustains 77.9 TFLOP/s with VGPR
// accumulators (tools/probe_mfma_f64_v2; AGPR accumulators run at less than half of that),
// but only if independent waves fill each other's waits: see the launcher for the measured
// tile / occupancy choice.
#include "common.h"
#include <cstdlib>

#ifndef MAUS_WC
#define MAUS_WC 8
#endif

namespace {

// TILED (LU workspace, luws.h): A, B and C are whole matrices stored tile-major -- tiles of 64 columns, the `lda` (= ldb = ldc)
// rows of a tile contiguous at 64 elements each -- and the operands are the sub-blocks
//   A[a_rows[m]][tcol.x + k],  B[tcol.y + k][tcol.z + n],  C[c_rows[m]][tcol.z + n]        (tcol.x, .y multiples of 16).
// A K-tile of 16 columns and a 16-column C block never straddle a tile, so inside a tile everything is row-major with a
// leading dimension of 64 and only the tile base moves.
struct TCol { int x, y, z; };
__device__ __forceinline__ long tile_off(long rows, int col) { return ((long)(col >> 6) * rows << 6) + (col & 63); }

template <int BM, int BN, int BK, int WM, int WN, int BLAY, bool CONJA, bool CONJB, bool PIPE, int MINW, bool M3 = false, bool TILED = false>
__global__ void __launch_bounds__(64 * WM * WN, MINW)
zgemm_kernel(int M, int N, int K,
             const c128* __restrict__ Ag, long lda, long strideA,
             const c128* __restrict__ Bg, long ldb, long strideB,
             c128* __restrict__ Cg, long ldc, long strideC,
             double alpha, int beta, int tiles_n, int nwg,
             const int* __restrict__ a_rows, const int* __restrict__ c_rows, long rows_stride, TCol tcol)
{
    static_assert(!TILED || (BLAY == 0 && !CONJA && !CONJB && BK == 16), "tiled operands: plain layout, K-tiles of 16");
    constexpr int NT = 64 * WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN;      // wave tile
    constexpr int MB = WTM / 16, NB = WTN / 16;      // 16x16 blocks per wave
    // LDS images are [k][m] / [k][n] with UNPADDED rows and the element index XOR-swizzled by (k & 7).
    // ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (banks mod 64
    // dwords): a fragment read (lanes 0-15 -> 16 consecutive elements of row k, lanes 16-31 -> row k+1) is
    // conflict-free exactly when consecutive rows have the same bank alignment, i.e. without padding --
    // the +1 padding used before cost 33 % extra LDS cycles (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).
    // ds_write_b128 is served in groups of 8 consecutive lanes (banks mod 32 dwords): the transposing
    // A store (8 lanes = 8 consecutive k of one row) lands on 8 different 16-B columns through the XOR,
    // which permutes inside aligned blocks of 8 and therefore keeps the reads conflict-free.
    constexpr int LDA_S = BM, LDB_S = BN;            // LDS row strides (elements)
    constexpr int A_PER = BM * BK / NT, B_PER = BN * BK / NT;
    static_assert(A_PER * NT == BM * BK && B_PER * NT == BN * BK, "tile/threads mismatch");

    constexpr int TILE_S = BK * LDA_S + BK * LDB_S;          // one staged K-tile (A then B), elements
    __shared__ c128 smem[(PIPE ? 2 : 1) * TILE_S];            // PIPE: double buffered, one barrier per K-tile

    // XCD-aware block -> tile map: blocks b and b+8 share an XCD (L2); give each XCD a
    // contiguous run of tiles so neighbouring tiles (same A row-panel) hit one L2.
    int bid = blockIdx.x;
    {
        int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // L2 blocking: tiles are walked column-block-major (WC tile columns wide, all tile rows), so the
    // B block of the current column block (WC x BN x K) stays resident in the XCD's 4 MB L2 while the
    // A row panels stream past it once.  Row-major order re-streamed all of B for every tile row
    // (PMC: FETCH_SIZE 3.5x the algorithmic reads; profiles/r01_pmc_traffic_before_l2_blocking.txt).
    // (4 / 8 / 16 / 32 tile columns per block: 79.3-79.8 TFLOP/s at K = 512 in every case, tools/gemm_k512_check.py)
#ifndef MAUS_WC
#define MAUS_WC 8
#endif
    constexpr int WC = MAUS_WC;
    const int tiles_m = nwg / tiles_n;
    const int full = tiles_m * WC;
    int cb = bid / full, rem = bid - cb * full, wl = WC;
    const int ncb = (tiles_n + WC - 1) / WC;
    if (cb >= ncb - 1) { cb = ncb - 1; rem = bid - cb * full; wl = tiles_n - cb * WC; }
    const int tm = rem / wl, tn = cb * WC + (rem - tm * wl);
    const int m0 = tm * BM, n0 = tn * BN;
    const long batch = blockIdx.y;
    const c128* A = Ag + batch * strideA;
    const c128* B = Bg + batch * strideB;
    c128* C = Cg + batch * strideC;
    // row gather (A) / scatter (C): one index list for every matrix (rows_stride = 0: population slots, Krylov rows) or
    // one list per matrix (the LU's row permutation: implicit pivoting)
    if (a_rows) a_rows += batch * rows_stride;
    if (c_rows) c_rows += batch * rows_stride;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;

    // 4M: cre / cim are the two result planes.  3M: cre = sum Are*Bre, cim = sum Aim*Bim,
    // c3 = sum (Are+Aim)*(Bre+Bim); combined in the epilogue.
    d4 cre[MB][NB], cim[MB][NB], c3[M3 ? MB : 1][M3 ? NB : 1];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            cre[i][j] = (d4){0, 0, 0, 0}; cim[i][j] = (d4){0, 0, 0, 0};
            if (M3) c3[i][j] = (d4){0, 0, 0, 0};
        }

    c128 ra[A_PER], rb[B_PER];

    // Branch-free tile loads: per-thread base pointers are computed once (row / column indices
    // clamped into range -- out-of-range rows and columns only feed C entries that are never
    // stored), so a K-tile costs one pointer add per element inside the loop.  Only when K is not
    // a multiple of BK (KEDGE) is the k index clamped and the value zeroed by a select.
    const bool KEDGE = (K % BK) != 0;
    const c128* pa[A_PER];
    const c128* pb[B_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        int r = tid / BK + i * (NT / BK);
        int gm = min(m0 + r, M - 1);
        pa[i] = A + (long)(a_rows ? a_rows[gm] : gm) * (TILED ? 64 : lda);
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        if (BLAY == 0) { int gn = min(n0 + tid % BN, N - 1); pb[i] = B + (TILED ? tile_off(ldb, tcol.z + gn) + (long)tcol.y * 64 : (long)gn); }
        else { int gn = min(n0 + tid / BK + i * (NT / BK), N - 1); pb[i] = B + (long)gn * ldb; }
    }
    const long ldb_e = TILED ? 64 : ldb;             // B row step inside the operand
    // plain layout: pb[i] points at this thread's B row for k0 = 0, so the per-tile step (k0 * ldb) is wave-uniform
    if (BLAY == 0) {
#pragma unroll
        for (int i = 0; i < B_PER; ++i) pb[i] += (long)(tid / BN + i * (NT / BN)) * ldb_e;
    }
    // The loads only LOAD: zeroing the K edge and conjugation happen when the registers are written to
    // LDS one K-tile later.  (Touching the values here makes the compiler wait for the loads right away,
    // which exposes the whole L2 latency in every K-tile: measured 76 -> 87 TFLOP/s on the 3M kernel.)
    int kload = 0;                                   // k0 of the tile held in ra / rb
    auto load_tiles = [&](int k0) {
        kload = k0;
        const long ta = TILED ? tile_off(lda, tcol.x + k0) : 0;          // wave-uniform: the K-tile's place in its 64-column tile
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int gk = k0 + (tid & (BK - 1));
            if (TILED) ra[i] = pa[i][ta + (tid & (BK - 1))];
            else ra[i] = pa[i][KEDGE ? min(gk, K - 1) : gk];
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            if (BLAY == 0) {
                const int kr = tid / BN + i * (NT / BN);
                if (!KEDGE || TILED) rb[i] = pb[i][(long)k0 * ldb_e];
                else rb[i] = pb[i][(long)(min(k0 + kr, K - 1) - kr) * ldb];
            } else {
                int gk = k0 + (tid & (BK - 1));
                rb[i] = pb[i][KEDGE ? min(gk, K - 1) : gk];
            }
        }
    };
    auto store_tiles = [&](int buf) {
        c128* As = smem + buf * TILE_S;
        c128* Bs = As + BK * LDA_S;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int k = tid & (BK - 1), r = tid / BK + i * (NT / BK);
            c128 v = ra[i];
            if (KEDGE && kload + k >= K) v = cmake(0.0, 0.0);
            if (CONJA) v.y = -v.y;
            As[k * LDA_S + (r ^ (k & 7))] = v;
        }
        if (BLAY == 0) {
#pragma unroll
            for (int i = 0; i < B_PER; ++i) {
                int n = tid % BN, k = tid / BN + i * (NT / BN);
                c128 v = rb[i];
                if (KEDGE && kload + k >= K) v = cmake(0.0, 0.0);
                if (CONJB) v.y = -v.y;
                Bs[k * LDB_S + (n ^ (k & 7))] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < B_PER; ++i) {
                int k = tid & (BK - 1), r = tid / BK + i * (NT / BK);
                c128 v = rb[i];
                if (KEDGE && kload + k >= K) v = cmake(0.0, 0.0);
                if (CONJB) v.y = -v.y;
                Bs[k * LDB_S + (r ^ (k & 7))] = v;
            }
        }
    };

    // ---- software-pipelined main loop ---------------------------------------------------------
    // LDS buffer t&1 holds K-tile t.  While the MFMAs of tile t run: the fragments of the next
    // k-step are already in flight (register double buffer), tile t+1 moves registers -> LDS, and
    // tile t+2 is requested from global memory.  One barrier per K-tile.
    const int nkt = (K + BK - 1) / BK;
    constexpr int KSTEPS = BK / 4;
    c128 fa[2][MB], fb[2][NB];
    auto read_frags = [&](int buf, int kk, int slot) {
        const c128* As = smem + buf * TILE_S;
        const c128* Bs = As + BK * LDA_S;
        const int krow = kk * 4 + (lane >> 4);
#pragma unroll
        for (int i = 0; i < MB; ++i) fa[slot][i] = As[krow * LDA_S + ((wm * WTM + i * 16 + (lane & 15)) ^ (krow & 7))];
#pragma unroll
        for (int j = 0; j < NB; ++j) fb[slot][j] = Bs[krow * LDB_S + ((wn * WTN + j * 16 + (lane & 15)) ^ (krow & 7))];
    };
    auto mfma_group = [&](int slot) {
        if (M3) {
            double as[MB], bs[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i) as[i] = fa[slot][i].x + fa[slot][i].y;
#pragma unroll
            for (int j = 0; j < NB; ++j) bs[j] = fb[slot][j].x + fb[slot][j].y;
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    cre[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[slot][i].x, fb[slot][j].x, cre[i][j], 0, 0, 0);
                    cim[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[slot][i].y, fb[slot][j].y, cim[i][j], 0, 0, 0);
                    c3[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[i], bs[j], c3[i][j], 0, 0, 0);
                }
        } else {
            // first products of every block (independent accumulators back to back) ...
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    cre[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[slot][i].x, fb[slot][j].x, cre[i][j], 0, 0, 0);
                    cim[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[slot][i].x, fb[slot][j].y, cim[i][j], 0, 0, 0);
                }
            // ... then the second products (BLGP=1: negate A -> Cre -= Aim*Bim)
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    cre[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[slot][i].y, fb[slot][j].y, cre[i][j], 0, 0, 1);
                    cim[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[slot][i].y, fb[slot][j].x, cim[i][j], 0, 0, 0);
                }
        }
    };

    if (PIPE) {
        load_tiles(0);
        store_tiles(0);
        if (nkt > 1) load_tiles(BK);                         // tile 1 in flight
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            const int buf = kt & 1;
            read_frags(buf, 0, 0);
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                if (kk + 1 < KSTEPS) read_frags(buf, kk + 1, (kk + 1) & 1);     // next k-step's fragments
                if (kk == 0 && kt + 1 < nkt) store_tiles(buf ^ 1);               // tile t+1: registers -> LDS
                if (kk == 0 && kt + 2 < nkt) load_tiles((kt + 2) * BK);          // tile t+2: global -> registers
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(kk & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
    } else {
        // lean form for >= 2 workgroups per CU: one LDS buffer, two barriers per K-tile; the bubbles of
        // one workgroup are filled by the independent waves of the others
        load_tiles(0);
        for (int kt = 0; kt < nkt; ++kt) {
            __syncthreads();
            store_tiles(0);
            __syncthreads();
            if (kt + 1 < nkt) load_tiles((kt + 1) * BK);
            read_frags(0, 0, 0);
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                if (kk + 1 < KSTEPS) read_frags(0, kk + 1, (kk + 1) & 1);
                mfma_group(kk & 1);
            }
        }
    }

    // epilogue: d[r] -> row (lane>>4) + 4r, col lane&15 of each 16x16 block; per block the four old
    // C values are fetched together, then combined and stored
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int gn = n0 + wn * WTN + j * 16 + (lane & 15);
            c128 cold[4];
            long off[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + wm * WTM + i * 16 + (lane >> 4) + 4 * r;
                const int cm = min(gm, M - 1), cn = min(gn, N - 1);
                off[r] = TILED ? (long)(c_rows ? c_rows[cm] : cm) * 64 + tile_off(ldc, tcol.z + cn)
                               : (long)(c_rows ? c_rows[cm] : cm) * ldc + cn;
                cold[r] = cmake(0.0, 0.0);
                if (beta) cold[r] = C[off[r]];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + wm * WTM + i * 16 + (lane >> 4) + 4 * r;
                if (gm < M && gn < N) {
                    const double vr = M3 ? cre[i][j][r] - cim[i][j][r] : cre[i][j][r];
                    const double vi = M3 ? (c3[i][j][r] - cre[i][j][r]) - cim[i][j][r] : cim[i][j][r];
                    C[off[r]] = cmake(alpha * vr + cold[r].x, alpha * vi + cold[r].y);
                }
            }
        }
}

// ---------------------------------------------------------------------------------------
// 3M zgemm with LDS-DMA staging (plain layout, the LU trailing updates).  Same 64 x 32 workgroup tile, 2 x 2 waves and
// 32 x 16 wave tile as the register-staged 3M kernel above, but the operand tiles go global -> LDS directly
// (global_load_lds_dwordx4: no staging VGPRs, no ds_write), in K-steps of 8 through a ring of NST buffers with ONE barrier
// per step.  Without the 24 staging VGPRs the kernel fits 96 VGPRs, i.e. five workgroups per CU instead of four.
//
// An LDS-DMA wave-instruction writes 64 x 16 B contiguously (lane-linear), so the LDS images are linear and any swizzle
// goes on the per-lane SOURCE address (and, identically, on the fragment read):
//   A image [64 rows][8 k]  (128 B per row): slot (r, kk) holds A[r][k0 + (kk ^ ((r >> 1) & 7))].  One instruction fills
//       8 rows (8 lanes = one row = 128 contiguous bytes of global memory).  The fragment read a = A[row = lane&15][k = lane>>4]
//       then takes 16 rows at a 128-B stride: the row parity selects the bank half and the XOR spreads the 8 rows of equal
//       parity over the 8 16-B columns -- conflict-free in every 16-lane group the hardware forms.
//   B image [8 k][32 n]  (512 B per k-row): natural order; the fragment read b = B[k = lane>>4][col = lane&15] is 16
//       consecutive elements of one row.
// ---------------------------------------------------------------------------------------
// BLAY = 1 (round 3): B is row-major [n][k] (dot-product form: Y = X A^T, Gram blocks, [V W][W V]^H) and gets the A image's
// treatment -- [BN rows][8 k], 8 lanes per row, the same source-side swizzle and fragment read.  CONJA / CONJB: the operand's
// imaginary part enters with the opposite sign, i.e. with a_i' = -Ai (b_i' = -Bi)
//     Re = Ar Br - a_i' b_i',   Im = (Ar + a_i')(Br + b_i') - Ar Br - a_i' b_i':
// the third product's operand sums become differences (one VALU op either way) and the epilogue takes P2 = Ai Bi with the sign
// sa sb; the MFMA stream is unchanged.  With these the population products (A@X of the Rayleigh / residual phases and of every
// GMRES inner iteration, A^H u of the SVD step, conj(X) V of the Hermitian match) run 6 M N K on the pipe instead of the 4M
// kernel's 8 M N K (MAUS_POPGEMM_3M=0: back to 4M).
template <int NST, int MINW, int MB, int NB, bool TILED = false, int BLAY = 0, bool CONJA = false, bool CONJB = false>
__global__ void __launch_bounds__(256, MINW)
zgemm3m_dma_kernel(int M, int N, int K,
                   const c128* __restrict__ Ag, long lda, long strideA,
                   const c128* __restrict__ Bg, long ldb, long strideB,
                   c128* __restrict__ Cg, long ldc, long strideC,
                   double alpha, int beta, int tiles_n, int nwg,
                   const int* __restrict__ a_rows, const int* __restrict__ c_rows, long rows_stride, TCol tcol)
{
    constexpr int WN = 2, BM = 2 * 16 * MB, BN = WN * 16 * NB, BKS = 8;      // 2 x 2 waves, wave tile (16 MB) x (16 NB)
    constexpr int A_ST = BM * BKS, B_ST = BKS * BN;            // elements per stage
    constexpr int A_PW = BM / 32;                              // A instructions per wave and stage (8 rows each)
    constexpr int B_PW = BN / 32;                              // B instructions per wave and stage (64 elements each)
    constexpr int DMA_PW = A_PW + B_PW;
    __shared__ c128 smem[NST * (A_ST + B_ST)];

    int bid = blockIdx.x;
    {
        int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int WC = MAUS_WC;
    const int tiles_m = nwg / tiles_n;
    const int full = tiles_m * WC;
    int cb = bid / full, rem = bid - cb * full, wl = WC;
    const int ncb = (tiles_n + WC - 1) / WC;
    if (cb >= ncb - 1) { cb = ncb - 1; rem = bid - cb * full; wl = tiles_n - cb * WC; }
    const int tm = rem / wl, tn = cb * WC + (rem - tm * wl);
    const int m0 = tm * BM, n0 = tn * BN;
    const long batch = blockIdx.y;
    const c128* A = Ag + batch * strideA;
    const c128* B = Bg + batch * strideB;
    c128* C = Cg + batch * strideC;
    if (a_rows) a_rows += batch * rows_stride;
    if (c_rows) c_rows += batch * rows_stride;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;

    // per-lane DMA sources.  A: wave w issues the 8-row blocks A_PW*w .. ; B: wave w issues the 64-element pieces B_PW*w ..
    const c128* srcA[A_PW];
#pragma unroll
    for (int j = 0; j < A_PW; ++j) {
        const int r = (A_PW * wave + j) * 8 + (lane >> 3);
        const int kk = (lane & 7) ^ ((r >> 1) & 7);
        const int gm = min(m0 + r, M - 1);
        srcA[j] = A + (long)(a_rows ? a_rows[gm] : gm) * (TILED ? 64 : lda) + kk;
    }
    static_assert(!(TILED && BLAY), "tiled operands: plain layout only");
    const c128* srcB[B_PW];
    const long ldb_e = TILED ? 64 : ldb;
#pragma unroll
    for (int j = 0; j < B_PW; ++j) {
        if (BLAY == 0) {
            const int e = (B_PW * wave + j) * 64 + lane;             // element of the [8][BN] image
            const int gn = min(n0 + (e % BN), N - 1);
            srcB[j] = B + (long)(e / BN) * ldb_e + (TILED ? tile_off(ldb, tcol.z + gn) + (long)tcol.y * 64 : (long)gn);
        } else {
            const int r = (B_PW * wave + j) * 8 + (lane >> 3);       // row of the [BN][8] image = column n of the product
            const int kk = (lane & 7) ^ ((r >> 1) & 7);
            srcB[j] = B + (long)min(n0 + r, N - 1) * ldb + kk;
        }
    }

    auto issue = [&](int st, int k0) {
        c128* As = smem + st * (A_ST + B_ST);
        c128* Bs = As + A_ST;
        const long ta = TILED ? tile_off(lda, tcol.x + k0) : (long)k0;     // wave-uniform; 8 consecutive k never leave a tile
#pragma unroll
        for (int j = 0; j < A_PW; ++j)
            __builtin_amdgcn_global_load_lds((const void*)(srcA[j] + ta), (__attribute__((address_space(3))) void*)(As + (A_PW * wave + j) * 64), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < B_PW; ++j)
            __builtin_amdgcn_global_load_lds((const void*)(srcB[j] + (BLAY ? (long)k0 : (long)k0 * ldb_e)), (__attribute__((address_space(3))) void*)(Bs + (B_PW * wave + j) * 64), 16, 0, 0);
    };

    d4 cre[MB][NB], cim[MB][NB], c3[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) { cre[i][j] = (d4){0, 0, 0, 0}; cim[i][j] = (d4){0, 0, 0, 0}; c3[i][j] = (d4){0, 0, 0, 0}; }

    const int nst = K / BKS;                         // K is a multiple of 8 on this path (launcher)
    // prologue: NST-1 stages in flight
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) if (s < nst) issue(s, s * BKS);

    const int arow0 = wm * 16 * MB + (lane & 15);
    const int bcol0 = wn * 16 * NB + (lane & 15);
    const int q = lane >> 4;
    for (int t = 0; t < nst; ++t) {
        // stage t has landed once at most NST-2 younger stages (DMA_PW instructions each) are still in flight
        if (NST >= 3 && t + NST - 2 < nst) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 2) * DMA_PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                // everybody's part of stage t is in LDS; everybody has left stage t-1
        asm volatile("" ::: "memory");
        const c128* As = smem + (t % NST) * (A_ST + B_ST);
        const c128* Bs = As + A_ST;
        // all fragment reads of the stage BEFORE the prefetch is issued: hipcc protects the first ds_read that follows a
        // global_load_lds it has seen with s_waitcnt vmcnt(0) (possible alias); issued first, the prefetch was waited for
        // at once (seen in the .s).  Behind the reads, the next ds_read is the one after the next rendezvous.
        c128 fa[2][MB], fb[2][NB];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int k = ks * 4 + q;
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const int r = arow0 + i * 16;
                fa[ks][i] = As[r * BKS + (k ^ ((r >> 1) & 7))];
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int cn = bcol0 + j * 16;
                fb[ks][j] = BLAY ? Bs[cn * BKS + (k ^ ((cn >> 1) & 7))] : Bs[k * BN + cn];
            }
        }
        if (t + NST - 1 < nst) issue((t + NST - 1) % NST, (t + NST - 1) * BKS);     // into the buffer stage t-1 used
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            double as[MB], bs[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i) as[i] = CONJA ? fa[ks][i].x - fa[ks][i].y : fa[ks][i].x + fa[ks][i].y;
#pragma unroll
            for (int j = 0; j < NB; ++j) bs[j] = CONJB ? fb[ks][j].x - fb[ks][j].y : fb[ks][j].x + fb[ks][j].y;
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    cre[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[ks][i].x, fb[ks][j].x, cre[i][j], 0, 0, 0);
                    cim[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[ks][i].y, fb[ks][j].y, cim[i][j], 0, 0, 0);
                    c3[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[i], bs[j], c3[i][j], 0, 0, 0);
                }
        }
    }
    constexpr double S2 = (CONJA != CONJB) ? -1.0 : 1.0;       // sign of P2 = Ai Bi in the result (sa sb)

    // epilogue (as in zgemm_kernel)
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int gn = n0 + wn * 16 * NB + j * 16 + (lane & 15);
            c128 cold[4];
            long off[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + wm * 16 * MB + i * 16 + (lane >> 4) + 4 * r;
                const int cm = min(gm, M - 1), cn = min(gn, N - 1);
                off[r] = TILED ? (long)(c_rows ? c_rows[cm] : cm) * 64 + tile_off(ldc, tcol.z + cn)
                               : (long)(c_rows ? c_rows[cm] : cm) * ldc + cn;
                cold[r] = cmake(0.0, 0.0);
                if (beta) cold[r] = C[off[r]];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + wm * 16 * MB + i * 16 + (lane >> 4) + 4 * r;
                if (gm < M && gn < N) {
                    const double vr = cre[i][j][r] - S2 * cim[i][j][r];
                    const double vi = (c3[i][j][r] - cre[i][j][r]) - S2 * cim[i][j][r];
                    C[off[r]] = cmake(alpha * vr + cold[r].x, alpha * vi + cold[r].y);
                }
            }
        }
}

template <int NST, int MINW, int MB, int NB, bool TILED = false, int BLAY = 0, bool CONJA = false, bool CONJB = false>
void launch_dma(hipStream_t st, int M, int N, int K, const c128* A, long lda, long sA, const c128* B, long ldb, long sB,
                c128* C, long ldc, long sC, double alpha, int beta, int batch, int, bool, bool,
                const int* a_rows, const int* c_rows, long rows_stride, TCol tcol = TCol{0, 0, 0})
{
    constexpr int BM = 32 * MB, BN = 32 * NB;
    int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    int nwg = tiles_m * tiles_n;
    hipLaunchKernelGGL((zgemm3m_dma_kernel<NST, MINW, MB, NB, TILED, BLAY, CONJA, CONJB>), dim3(nwg, batch), dim3(256), 0, st,
                       M, N, K, A, lda, sA, B, ldb, sB, C, ldc, sC, alpha, beta, tiles_n, nwg, a_rows, c_rows, rows_stride, tcol);
}

// population products (dot-product layout and / or conjugated operands) on the DMA-staged 3M kernel: 64 x 64 tiles for large
// products, 64 x 32 otherwise
template <int BLAY, bool CONJA, bool CONJB>
void launch_dma_pop(hipStream_t st, int M, int N, int K, const c128* A, long lda, long sA, const c128* B, long ldb, long sB,
                    c128* C, long ldc, long sC, double alpha, int beta, int batch, int blay, bool conja, bool conjb,
                    const int* a_rows, const int* c_rows, long rows_stride)
{
    if (M >= 1536 && N >= 1536) launch_dma<2, 3, 2, 2, false, BLAY, CONJA, CONJB>(st, M, N, K, A, lda, sA, B, ldb, sB, C, ldc, sC, alpha, beta, batch, blay, conja, conjb, a_rows, c_rows, rows_stride);
    else launch_dma<2, 5, 2, 1, false, BLAY, CONJA, CONJB>(st, M, N, K, A, lda, sA, B, ldb, sB, C, ldc, sC, alpha, beta, batch, blay, conja, conjb, a_rows, c_rows, rows_stride);
}

template <int BM, int BN, int BK, int WM, int WN, bool PIPE, int MINW>
void launch_cfg(hipStream_t st, int M, int N, int K, const c128* A, long lda, long sA, const c128* B, long ldb, long sB,
                c128* C, long ldc, long sC, double alpha, int beta, int batch, int blay, bool conja, bool conjb,
                const int* a_rows, const int* c_rows, long rows_stride)
{
    int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    int nwg = tiles_m * tiles_n;
    dim3 grid(nwg, batch), block(64 * WM * WN);
#define LAUNCH(BL, CA, CB) hipLaunchKernelGGL((zgemm_kernel<BM, BN, BK, WM, WN, BL, CA, CB, PIPE, MINW>), grid, block, 0, st, \
        M, N, K, A, lda, sA, B, ldb, sB, C, ldc, sC, alpha, beta, tiles_n, nwg, a_rows, c_rows, rows_stride, TCol{0, 0, 0})
    if (blay == 0) {
        if (!conja && !conjb) LAUNCH(0, false, false);
        else if (!conja && conjb) LAUNCH(0, false, true);
        else if (conja && !conjb) LAUNCH(0, true, false);
        else LAUNCH(0, true, true);
    } else {
        if (!conja && !conjb) LAUNCH(1, false, false);
        else if (!conja && conjb) LAUNCH(1, false, true);
        else if (conja && !conjb) LAUNCH(1, true, false);
        else LAUNCH(1, true, true);
    }
#undef LAUNCH
}

// plain-layout-only instantiation (LU trailing updates): one kernel per tile shape instead of eight
template <int BM, int BN, int BK, int WM, int WN, bool PIPE = false, int MINW = 4, bool M3 = false, bool TILED = false>
void launch_lu_only(hipStream_t st, int M, int N, int K, const c128* A, long lda, long sA, const c128* B, long ldb, long sB,
                    c128* C, long ldc, long sC, double alpha, int beta, int batch, int, bool, bool,
                    const int* a_rows, const int* c_rows, long rows_stride, TCol tcol = TCol{0, 0, 0})
{
    int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    int nwg = tiles_m * tiles_n;
    hipLaunchKernelGGL((zgemm_kernel<BM, BN, BK, WM, WN, 0, false, false, PIPE, MINW, M3, TILED>), dim3(nwg, batch), dim3(64 * WM * WN), 0, st,
                       M, N, K, A, lda, sA, B, ldb, sB, C, ldc, sC, alpha, beta, tiles_n, nwg, a_rows, c_rows, rows_stride, tcol);
}

}  // namespace

// Host-side launcher (device pointers).  batch matrices at element strides sA/sB/sC.
// a_rows / c_rows (device int arrays of length M, or null): row gather for A / row scatter
// for C -- the population's candidate vectors live in arbitrary slots of one array.
void maus_zgemm_launch_rows(hipStream_t st, int M, int N, int K, const c128* A, long lda, long sA,
                            const c128* B, long ldb, long sB, c128* C, long ldc, long sC,
                            double alpha, int beta, int batch, int blay, bool conja, bool conjb,
                            const int* a_rows, const int* c_rows, long rows_stride)
{
    if (M <= 0 || N <= 0 || batch <= 0) return;
#define ARGS st, M, N, K, A, lda, sA, B, ldb, sB, C, ldc, sC, alpha, beta, batch, blay, conja, conjb, a_rows, c_rows, rows_stride
    // All kernels: 4 waves per workgroup and registers / LDS small enough for 3-5 INDEPENDENT workgroups per CU.  Measured on
    // MI355X in rounds 1-3 (profiles/r01_gemm_sweep_configs.txt, r02_zgemm_tile_variants_small_shapes.txt; K = 256, 136 matrices,
    // 8MNK-equivalent TFLOP/s): every one-workgroup-per-CU shape (128 x 64 / 128 x 128, BK 16 / 32, software-pipelined or not)
    // stays at 50-59 -- with all waves of a SIMD in one workgroup they run in lockstep and every wait or barrier of one is a
    // bubble for all; 8-wave workgroups lose to 4-wave ones at equal tile area.  The variants that lost those sweeps are gone.
    constexpr int DMA_KMIN = 64;
    if (blay == 0 && !conja && !conjb) {
        // plain layout (row-major LU workspaces of the GMRES path, host-matrix entry points).  Skinny shapes keep the workgroup
        // tile shaped like the problem so that no MFMA runs on padding.
        if (N <= 16) { launch_lu_only<128, 16, 16, 4, 1>(ARGS); return; }
        if (M <= 16) { launch_lu_only<16, 128, 16, 1, 4, false, 3>(ARGS); return; }
        // K >= 64: the LDS-DMA staged 3M kernel.  3584 x 3616 x 512, 136 matrices: register-staged kernel 79.2; DMA staging,
        // 64 x 32 tiles, five workgroups per CU 86.8 (prefetch issued behind the fragment reads); 64 x 64 tiles at three
        // workgroups per CU 88.7 on large updates (86.3 at 1024 x 1056: the 64 x 32 form is kept below 1536).  Same summation
        // order as the register-staged kernel, hence the same bits.
        if (M > 32 && (K % 8) == 0 && K >= DMA_KMIN) {
            if (M >= 1536 && N >= 1536) launch_dma<2, 3, 2, 2>(ARGS);   // 64 x 64 tiles, three workgroups per CU
            else launch_dma<2, 5, 2, 1>(ARGS);
            return;
        }
        // 3M complex product on the register-staged kernel: three real MFMA products per complex one (ArBr, AiBi,
        // (Ar+Ai)(Br+Bi)); a 32 x 16 wave tile keeps the three accumulator planes, the fragments and the in-flight prefetch of
        // the next K-tile inside 128 VGPRs.  Error is normwise the same as 4M (9.5e-16 vs 1.1e-15 relative on random data); the
        // imaginary part loses its componentwise bound, which LU with partial pivoting does not rely on.
        if (M <= 32) { launch_lu_only<32, 64, 16, 1, 4, false, 4, true>(ARGS); return; }
        launch_lu_only<64, 32, 16, 2, 2, false, 4, true>(ARGS); return;
    }
    // Population products with a dot-product B layout and / or conjugated operands: the DMA-staged 3M kernel (round 3) where
    // its staging applies (K a multiple of 8, >= 64, more than one 32-row tile)
    if (M > 32 && (K % 8) == 0 && K >= DMA_KMIN) {
        if (blay == 1 && !conja && !conjb) { launch_dma_pop<1, false, false>(ARGS); return; }
        if (blay == 1 && conja && !conjb) { launch_dma_pop<1, true, false>(ARGS); return; }
        if (blay == 1 && !conja && conjb) { launch_dma_pop<1, false, true>(ARGS); return; }
        if (blay == 0 && conja && !conjb) { launch_dma_pop<0, true, false>(ARGS); return; }
        if (blay == 0 && !conja && conjb) { launch_dma_pop<0, false, true>(ARGS); return; }
    }
    // 4M otherwise (a handful of candidates, K not a multiple of 8): 64 x 64 tiles, 32 x 32 wave tile; with the next K-tile
    // genuinely in flight during the MFMAs this needs ~160 VGPRs: three workgroups per CU.  A population of a few hundred
    // candidates against an n x n matrix gives few such tiles (M = 512, N = 2048: 256 -- one workgroup on each CU where three
    // fit): below two tiles per CU the tile shrinks to 32 x 64 / 32 x 32 so that the grid covers the chip (M = 15 against
    // 8192 x 8192 runs 0.57 ms per product on 256 workgroups of 32 x 32, 0.86 on 128 of 64 x 64).
    const long t64 = (long)((M + 63) / 64) * ((N + 63) / 64) * batch;
    const int tile = (t64 >= 512) ? 0 : ((long)((M + 31) / 32) * ((N + 63) / 64) * batch >= 512 ? 1 : 2);
    if (tile == 1) { launch_cfg<32, 64, 16, 1, 4, false, 4>(ARGS); return; }
    if (tile == 2) { launch_cfg<32, 32, 16, 2, 2, false, 4>(ARGS); return; }
    launch_cfg<64, 64, 16, 2, 2, false, 3>(ARGS);
#undef ARGS
}

// LU trailing update on the tile-major workspace (luws.h):  H[rows[m]][ccol + n] -= H[rows[m]][acol + k] * U[brow + k][ccol + n]
// for m < M, n < N, k < K, all `batch` matrices (element stride `stride`, `nrows` rows per 64-column tile); rows = the LU's
// per-matrix row permutation (rows_stride apart).  Same kernels and dispatch rule as the plain-layout path above.
void maus_zgemm_launch_lu(hipStream_t st, int M, int N, int K, const c128* H, const c128* U, c128* Hc, long nrows, long stride,
                          int acol, int brow, int ccol, int batch, const int* rows, long rows_stride)
{
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return;
    constexpr int dma_kmin = 64;
    const TCol tc{acol, brow, ccol};
#define ARGS st, M, N, K, H, nrows, stride, U, nrows, stride, Hc, nrows, stride, -1.0, 1, batch, 0, false, false, rows, rows, rows_stride, tc
    if (N <= 16) { launch_lu_only<128, 16, 16, 4, 1, false, 4, false, true>(ARGS); return; }
    if (M <= 16) { launch_lu_only<16, 128, 16, 1, 4, false, 3, false, true>(ARGS); return; }
    if (M > 32 && (K % 8) == 0 && K >= dma_kmin) {
        // (round 3: 64 x 64 tiles from M >= 1024 / 512 and N >= 128 / 256 / 1024, or only from 2048: the 181-solve sweep stays
        // within 0.5 ms of 494 ms -- the K = 128 / 256 levels are not bound by the tile shape)
        if (M >= 1536 && N >= 1536) launch_dma<2, 3, 2, 2, true>(ARGS);      // 64 x 64 tiles, three workgroups per CU
        else launch_dma<2, 5, 2, 1, true>(ARGS);
        return;
    }
    if (M <= 32) { launch_lu_only<32, 64, 16, 1, 4, false, 4, true, true>(ARGS); return; }
    launch_lu_only<64, 32, 16, 2, 2, false, 4, true, true>(ARGS);
#undef ARGS
}

void maus_zgemm_launch_idx(hipStream_t st, int M, int N, int K, const c128* A, long lda, long sA,
                           const c128* B, long ldb, long sB, c128* C, long ldc, long sC,
                           double alpha, int beta, int batch, int blay, bool conja, bool conjb,
                           const int* a_rows, const int* c_rows)
{
    maus_zgemm_launch_rows(st, M, N, K, A, lda, sA, B, ldb, sB, C, ldc, sC, alpha, beta, batch, blay, conja, conjb, a_rows, c_rows, 0);
}

void maus_zgemm_launch(hipStream_t st, int M, int N, int K, const c128* A, long lda, long sA,
                       const c128* B, long ldb, long sB, c128* C, long ldc, long sC,
                       double alpha, int beta, int batch, int blay, bool conja, bool conjb)
{
    maus_zgemm_launch_idx(st, M, N, K, A, lda, sA, B, ldb, sB, C, ldc, sC, alpha, beta, batch, blay, conja, conjb, nullptr, nullptr);
}
