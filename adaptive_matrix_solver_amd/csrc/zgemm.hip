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
// (k-major, +1 element row padding) so every fragment is one ds_read_b128 of an
// interleaved (re,im) pair; the next K-tile is prefetched into registers while the
// current one feeds the MFMAs.  >= 2 waves per SIMD are needed to keep the fp64
// matrix pipe issuing back to back (probe: 35 TF at 1 wave/SIMD, 47 TF at 2).
#include "common.h"

namespace {

constexpr int BK = 16;

template <int BM, int BN, int WM, int WN, int BLAY, bool CONJA, bool CONJB>
__global__ void __launch_bounds__(64 * WM * WN)
zgemm_kernel(int M, int N, int K,
             const c128* __restrict__ Ag, long lda, long strideA,
             const c128* __restrict__ Bg, long ldb, long strideB,
             c128* __restrict__ Cg, long ldc, long strideC,
             double alpha, int beta, int tiles_n, int nwg,
             const int* __restrict__ a_rows, const int* __restrict__ c_rows)
{
    constexpr int NT = 64 * WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN;      // wave tile
    constexpr int MB = WTM / 16, NB = WTN / 16;      // 16x16 blocks per wave
    constexpr int LDA_S = BM + 1, LDB_S = BN + 1;    // LDS row strides (elements)
    constexpr int A_PER = BM * BK / NT, B_PER = BN * BK / NT;
    static_assert(A_PER * NT == BM * BK && B_PER * NT == BN * BK, "tile/threads mismatch");

    __shared__ c128 smem[BK * LDA_S + BK * LDB_S];
    c128* As = smem;
    c128* Bs = smem + BK * LDA_S;

    // XCD-aware block -> tile map: blocks b and b+8 share an XCD (L2); give each XCD a
    // contiguous run of tiles so neighbouring tiles (same A row-panel) hit one L2.
    int bid = blockIdx.x;
    {
        int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const long batch = blockIdx.y;
    const c128* A = Ag + batch * strideA;
    const c128* B = Bg + batch * strideB;
    c128* C = Cg + batch * strideC;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;

    d4 cre[MB][NB], cim[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) { cre[i][j] = (d4){0, 0, 0, 0}; cim[i][j] = (d4){0, 0, 0, 0}; }

    c128 ra[A_PER], rb[B_PER];

    auto load_tiles = [&](int k0) {
        // A tile: [BM rows][BK k], k contiguous in memory
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int k = tid & (BK - 1), r = (tid >> 4) + i * (NT / BK);
            int gm = m0 + r, gk = k0 + k;
            c128 v = cmake(0.0, 0.0);
            if (gm < M && gk < K) v = A[(long)(a_rows ? a_rows[gm] : gm) * lda + gk];
            if (CONJA) v.y = -v.y;
            ra[i] = v;
        }
        if (BLAY == 0) {   // B[k][n], n contiguous
#pragma unroll
            for (int i = 0; i < B_PER; ++i) {
                int n = tid % BN, k = tid / BN + i * (NT / BN);
                int gn = n0 + n, gk = k0 + k;
                c128 v = cmake(0.0, 0.0);
                if (gn < N && gk < K) v = B[(long)gk * ldb + gn];
                if (CONJB) v.y = -v.y;
                rb[i] = v;
            }
        } else {           // B[n][k], k contiguous
#pragma unroll
            for (int i = 0; i < B_PER; ++i) {
                int k = tid & (BK - 1), r = (tid >> 4) + i * (NT / BK);
                int gn = n0 + r, gk = k0 + k;
                c128 v = cmake(0.0, 0.0);
                if (gn < N && gk < K) v = B[(long)gn * ldb + gk];
                if (CONJB) v.y = -v.y;
                rb[i] = v;
            }
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int k = tid & (BK - 1), r = (tid >> 4) + i * (NT / BK);
            As[k * LDA_S + r] = ra[i];
        }
        if (BLAY == 0) {
#pragma unroll
            for (int i = 0; i < B_PER; ++i) {
                int n = tid % BN, k = tid / BN + i * (NT / BN);
                Bs[k * LDB_S + n] = rb[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < B_PER; ++i) {
                int k = tid & (BK - 1), r = (tid >> 4) + i * (NT / BK);
                Bs[k * LDB_S + r] = rb[i];
            }
        }
    };

    const int nkt = (K + BK - 1) / BK;
    load_tiles(0);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();            // everyone finished reading the previous tile
        store_tiles();
        __syncthreads();
        if (kt + 1 < nkt) load_tiles((kt + 1) * BK);   // in flight during the MFMAs below
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            const int krow = kk * 4 + (lane >> 4);
            c128 a[MB], b[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i) a[i] = As[krow * LDA_S + wm * WTM + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < NB; ++j) b[j] = Bs[krow * LDB_S + wn * WTN + j * 16 + (lane & 15)];
            // first products of every block (independent accumulators back to back) ...
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    cre[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[j].x, cre[i][j], 0, 0, 0);
                    cim[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[j].y, cim[i][j], 0, 0, 0);
                }
            // ... then the second products (BLGP=1: negate A -> Cre -= Aim*Bim)
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    cre[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[j].y, cre[i][j], 0, 0, 1);
                    cim[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[j].x, cim[i][j], 0, 0, 0);
                }
        }
    }

    // epilogue: d[r] -> row (lane>>4) + 4r, col lane&15 of each 16x16 block
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int gn = n0 + wn * WTN + j * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + wm * WTM + i * 16 + (lane >> 4) + 4 * r;
                if (gm < M && gn < N) {
                    c128* p = C + (long)(c_rows ? c_rows[gm] : gm) * ldc + gn;
                    c128 v = cmake(alpha * cre[i][j][r], alpha * cim[i][j][r]);
                    if (beta) { c128 o = *p; v.x += o.x; v.y += o.y; }
                    *p = v;
                }
            }
        }
}

template <int BM, int BN, int WM, int WN>
void launch_cfg(hipStream_t st, int M, int N, int K, const c128* A, long lda, long sA, const c128* B, long ldb, long sB,
                c128* C, long ldc, long sC, double alpha, int beta, int batch, int blay, bool conja, bool conjb,
                const int* a_rows, const int* c_rows)
{
    int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    int nwg = tiles_m * tiles_n;
    dim3 grid(nwg, batch), block(64 * WM * WN);
#define LAUNCH(BL, CA, CB) hipLaunchKernelGGL((zgemm_kernel<BM, BN, WM, WN, BL, CA, CB>), grid, block, 0, st, \
        M, N, K, A, lda, sA, B, ldb, sB, C, ldc, sC, alpha, beta, tiles_n, nwg, a_rows, c_rows)
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

}  // namespace

// Host-side launcher (device pointers).  batch matrices at element strides sA/sB/sC.
// a_rows / c_rows (device int arrays of length M, or null): row gather for A / row scatter
// for C -- the population's candidate vectors live in arbitrary slots of one array.
void maus_zgemm_launch_idx(hipStream_t st, int M, int N, int K, const c128* A, long lda, long sA,
                           const c128* B, long ldb, long sB, c128* C, long ldc, long sC,
                           double alpha, int beta, int batch, int blay, bool conja, bool conjb,
                           const int* a_rows, const int* c_rows)
{
    if (M <= 0 || N <= 0 || batch <= 0) return;
    // 128x64 tiles (8 waves, 2/SIMD at one block per CU) once the problem fills the chip with
    // them; 64x64 (4 waves, two blocks per CU) otherwise.
    long t128 = (long)((M + 127) / 128) * ((N + 63) / 64) * batch;
    if (M >= 128 && t128 >= 512)
        launch_cfg<128, 64, 4, 2>(st, M, N, K, A, lda, sA, B, ldb, sB, C, ldc, sC, alpha, beta, batch, blay, conja, conjb, a_rows, c_rows);
    else
        launch_cfg<64, 64, 2, 2>(st, M, N, K, A, lda, sA, B, ldb, sB, C, ldc, sC, alpha, beta, batch, blay, conja, conjb, a_rows, c_rows);
}

void maus_zgemm_launch(hipStream_t st, int M, int N, int K, const c128* A, long lda, long sA,
                       const c128* B, long ldb, long sB, c128* C, long ldc, long sC,
                       double alpha, int beta, int batch, int blay, bool conja, bool conjb)
{
    maus_zgemm_launch_idx(st, M, N, K, A, lda, sA, B, ldb, sB, C, ldc, sC, alpha, beta, batch, blay, conja, conjb, nullptr, nullptr);
}
