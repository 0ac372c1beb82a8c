// Batched Psi-regularised shifted LU solve for the MAUS candidate step (gfx950).
//
// Replaces, per candidate k (AMS = the reference file):
//   A_t   = A - lambda_k * eye(N)                               AMS:270
//   reg   = psi_k * eye(N) + 0.15*psi_k*((U1-.5) + i(U2-.5))    AMS:49-50
//   H     = A_t + reg                                           AMS:52
//   w     = scipy.linalg.solve(H, rhs)  (zgetrf + zgetrs)       AMS:59
//
// Design (one workspace of G matrices, all kernels batched over G):
//   * H is row-major [npad][ldh], npad = roundup(n,32), ldh = npad + 32.  The pad block is the
//     identity (never pivoted into), and column `npad` carries the right-hand side, so the
//     forward substitution L y = P b happens inside the factorisation and only U x = y remains.
//   * right-looking outer blocks of NBO (512) columns; each block column is factored by a host-driven
//     recursion (halving down to 16-wide panels) so that every flop outside the base panels is a
//     call of the MFMA zgemm.  The L part to the left of a block is never needed again because y is
//     carried in the augmented column.
//   * implicit row pivoting: rows never move, every kernel addresses them through perm[] (see below);
//     finished U rows are collected in a second array in logical order.
//   * base panel (16 columns): one workgroup per matrix (several for small batches), rows owned by
//     threads, 4- or 2-column sub-blocks kept in registers and brought up to date left-looking while
//     they are loaded; pivot rule = LAPACK izamax (max |re|+|im|, first index wins).
#include "common.h"
#include "luws.h"
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <algorithm>
#include <type_traits>

void maus_zgemm_launch(hipStream_t st, int M, int N, int K, const c128* A, long lda, long sA,
                       const c128* B, long ldb, long sB, c128* C, long ldc, long sC,
                       double alpha, int beta, int batch, int blay, bool conja, bool conjb);

void maus_zgemm_launch_lu(hipStream_t st, int M, int N, int K, const c128* H, const c128* U, c128* Hc, long nrows, long stride,
                          int acol, int brow, int ccol, int batch, const int* rows, long rows_stride);

void maus_zgemm_launch_rows(hipStream_t st, int M, int N, int K, const c128* A, long lda, long sA,
                            const c128* B, long ldb, long sB, c128* C, long ldc, long sC,
                            double alpha, int beta, int batch, int blay, bool conja, bool conjb,
                            const int* a_rows, const int* c_rows, long rows_stride);

namespace {

#ifndef MAUS_NBP
#define MAUS_NBP 16
#endif
constexpr int NBP = MAUS_NBP;   // base panel width.  16 vs 32 with the left-looking panel: 294-297 vs 296 candidate-steps/s
                                // (the wider panel costs what the K=16 update level saves); with the right-looking panel 16 won clearly
constexpr int AUG = 32;         // augmented (rhs) column block, also the padding granule of n
constexpr int BSB = 32;         // back-substitution block
constexpr int PW = 4;       // register sub-block width inside the panel
constexpr int PT = 512;     // panel threads (8 waves, 2 per SIMD -> 256 VGPR budget)

// ---------------------------------------------------------------------------------------
// H build.  grid = (npad, G), block = 256.  pert_mode 0: none, 1: uniform draws supplied.
// Rounding order mirrors NumPy exactly: (a - lam*delta) + (psi*delta + ((u-.5)*psi)*0.15).
// flags[g] bit0 <- any non-finite entry of H or rhs (scipy's check_finite -> ValueError).
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
build_h_kernel(const c128* __restrict__ A, int n, int npad, long ldh, long strideH, c128* __restrict__ Hg,
               const c128* __restrict__ shift, const double* __restrict__ psi,
               int rhs_mode, const c128* __restrict__ X, long ldx, const int* __restrict__ slots,
               const c128* __restrict__ bvec,
               int pert_mode, const double* __restrict__ U /* [G][2][n][n] */,
               int* __restrict__ flags, int tiled)
{
    const int i = blockIdx.x, g = blockIdx.y;
    // element (i, j): row-major with leading dimension ldh (GMRES operand) or tile-major (LU workspace, luws.h)
    c128* Hmat = Hg + (long)g * strideH;
    auto at = [&](int j) -> c128& { return Hmat[tiled ? lu_tix(npad, i, j) : (long)i * ldh + j]; };
    const c128 lam = shift[g];
    const double ps = psi[g];
    bool bad = false;
    if (i < n) {
        const c128* Arow = A + (long)i * n;
        const double* U1 = (pert_mode == 1) ? U + ((long)g * 2 + 0) * n * n + (long)i * n : nullptr;
        const double* U2 = (pert_mode == 1) ? U + ((long)g * 2 + 1) * n * n + (long)i * n : nullptr;
        for (int j = threadIdx.x; j < n; j += blockDim.x) {
            c128 a = Arow[j];
            double pr = 0.0, pi = 0.0;
            if (pert_mode == 1) {
                pr = __dmul_rn(__dmul_rn(__dsub_rn(U1[j], 0.5), ps), 0.15);
                pi = __dmul_rn(__dmul_rn(__dsub_rn(U2[j], 0.5), ps), 0.15);
            }
            c128 h;
            if (j == i) {
                h.x = __dadd_rn(__dsub_rn(a.x, lam.x), __dadd_rn(ps, pr));
                h.y = __dadd_rn(__dsub_rn(a.y, lam.y), __dadd_rn(0.0, pi));
            } else {
                h.x = __dadd_rn(a.x, pr);
                h.y = __dadd_rn(a.y, pi);
            }
            bad |= !cfinite(h);
            at(j) = h;
        }
        for (int j = n + threadIdx.x; j < npad; j += blockDim.x) at(j) = cmake(0.0, 0.0);
    } else {
        for (int j = threadIdx.x; j < npad; j += blockDim.x) at(j) = cmake(j == i ? 1.0 : 0.0, 0.0);
    }
    // augmented block: column npad = rhs, the other 31 columns zero
    for (int j = threadIdx.x; j < AUG; j += blockDim.x) {
        c128 v = cmake(0.0, 0.0);
        if (j == 0 && i < n) {
            v = (rhs_mode == 0) ? X[(long)slots[g] * ldx + i] : bvec[i];
            bad |= !cfinite(v);
        }
        at(npad + j) = v;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&flags[g], 1);
}

// host-provided dense matrices (maus_lu_solve_host): H <- [A_g | b_g], padded
__global__ void __launch_bounds__(256)
load_h_kernel(const c128* __restrict__ Ain /*[G][n][n]*/, const c128* __restrict__ bin /*[G][n]*/, int n, int npad,
              long ldh, long strideH, c128* __restrict__ Hg, int* __restrict__ flags)
{
    const int i = blockIdx.x, g = blockIdx.y;
    c128* Hmat = Hg + (long)g * strideH;                         // tile-major (luws.h)
    (void)ldh;
    bool bad = false;
    for (int j = threadIdx.x; j < npad; j += blockDim.x) {
        c128 v = cmake((i == j && i >= n) ? 1.0 : 0.0, 0.0);
        if (i < n && j < n) { v = Ain[((long)g * n + i) * n + j]; bad |= !cfinite(v); }
        Hmat[lu_tix(npad, i, j)] = v;
    }
    for (int j = threadIdx.x; j < AUG; j += blockDim.x) {
        c128 v = cmake(0.0, 0.0);
        if (j == 0 && i < n) { v = bin[(long)g * n + i]; bad |= !cfinite(v); }
        Hmat[lu_tix(npad, i, npad + j)] = v;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&flags[g], 1);
}

// =======================================================================================
// Implicit pivoting.  Rows never move: perm[i] is the physical row of H holding logical row i, a row
// interchange is a swap of two perm entries, and every kernel addresses rows through perm (the zgemm gathers its A / C
// rows through the same list).  That removes round 1's row-swap sweeps altogether -- 600 MB of HBM traffic per matrix and
// factorisation at n = 4096 (each outer block's 512 interchanges applied to everything to its right, plus the
// interchanges inside the block column at every recursion level), 5 % of the step -- and the in-panel swaps of the
// other 12 panel columns.  The pivot choice is unchanged: the search runs over LOGICAL row indices, first index wins,
// so ipiv is LAPACK's sequence.  Finished U rows go to the second array U in logical order (written once, by the
// triangular solves and the panel), which keeps the B operand of every update and the back substitution contiguous.
// =======================================================================================
__global__ void __launch_bounds__(256)
init_perm_kernel(int* __restrict__ perm_g, int npad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npad) perm_g[(long)blockIdx.y * npad + i] = i;
}

#ifdef MAUS_PANEL_CLOCK
__device__ unsigned long long g_panel_clk[16];
#define PCLK(i) do { if (tid == 0) { unsigned long long t_ = wall_clock64(); atomicAdd(&g_panel_clk[i], t_ - tclk); tclk = t_; } } while (0)
#define PCLK_SYNC(i) do { __syncthreads(); PCLK(i); } while (0)
#else
#define PCLK(i)
#define PCLK_SYNC(i)
#endif
// Base panel: LU with partial pivoting of the m x NBP block at (j0, j0), m = npad - j0.  One workgroup (512 threads) per
// matrix; thread t owns the logical rows t, t+512, ... (RPT of them) and keeps one PWL-column sub-block of them in
// registers.  Left-looking over the sub-blocks: a sub-block applies the updates of the EARLIER sub-blocks to its own columns
// while loading them,
//     R[r][c] = A[r][c0+c] - sum_{j<c0} L[r][j] * U[j][c0+c]
// instead of every sub-block pushing a rank-PWL update through memory into the columns to its right: per 16-wide panel 40
// column reads + 16 column writes instead of 56 + 36, every row read as one contiguous piece of 64..256 bytes.
//   (b') U[0:c0][c0:c0+PWL] = L00^-1 A[pivot rows][c0:c0+PWL]   (unit-lower L00 of the finished sub-blocks, <= 12x12,
//        forward substitution by one lane per column out of LDS)
//   (c') the row update above, U broadcast from LDS, L read from the row's own finished columns
//   (d') the pivot steps on the register sub-block: max |re|+|im| over the logical rows >= the diagonal, first index wins
//        (LAPACK izamax); the two owner threads exchange register rows and the physical rows they stand for
// PWL = 4 columns with up to 8 rows per thread (m <= 4096), 2 columns with 16 rows per thread (m <= 8192) -- the same 128
// VGPRs of sub-block either way.
// Where its time goes (in-kernel clocks, -DMAUS_PANEL_CLOCK + tools/panel_clocks.py, profiles/r02_panel_phase_clocks.txt):
// per panel ~45 us (32 solves) to ~100 us (181 solves) in the left-looking loads, ~38 us in the 16 column steps (2.4 us
// each, instruction-bound: ~800 instructions per step and wave at 4 rows per thread), ~20 us store + barrier, ~9 us (b').
// One workgroup per matrix draws 20-30 GB/s -- a CU's share of the chip's bandwidth -- whatever the access pattern.
// Round 3: requesting the next sub-block's raw columns right behind the stores (they do not depend on them) so that they
// arrive while the barrier waits for the stores to drain: 35.3 vs 32.1 ms per 181-solve sweep, slower, not adopted.
// Measured and rejected in round 2 (each bit-identical to this kernel, none faster at 32 or 181 solves per call):
// a column-major scratch copy of the panel (coalesced sub-block loads, but the two transpositions cost what they save:
// 51.5 vs 53.6 ms per 181-solve sweep, 34.2 vs 30.0 at 32); 1024 threads per workgroup (69 vs 54 ms); one barrier per
// column with per-wave candidate rows published before it, DPP maximum and pivot rows kept in LDS (67 vs 54 ms: the
// column step is bound by its instruction count, not by its barriers).
// The inverse of the unit lower triangular 16 x 16 diagonal block of the panel a workgroup has just factored (round 4: the
// triangular solves multiply by it, trsm_mfma_kernel).  sL = the block (only its strictly lower part is read); thread c < 16
// substitutes column c of the inverse -- x_c = 1, x_i = -sum_{k<i} L[i][k] x_k below it -- and stores its strictly lower part
// where nothing else lives: below the diagonal of the same block of the logical-order U array (Um = its row 0, column 0).
// Inside the panel kernels this costs one wave a few microseconds at the very end; as a kernel of its own (the first version)
// it was 256 launches of ~7 us per factorisation.
__device__ __forceinline__ void panel_store_diaginv(const c128 (*sL)[17], c128* __restrict__ Um, long ld, int tid)
{
    if (tid < 16) {
        c128 x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            c128 sacc = cmake(i == tid ? 1.0 : 0.0, 0.0);
            if (i > tid) {
#pragma unroll
                for (int k = 0; k < i; ++k) cfms(sacc, sL[i][k], x[k]);
            }
            x[i] = sacc;
        }
#pragma unroll
        for (int i = 1; i < 16; ++i) if (i > tid) Um[(long)i * ld + tid] = x[i];
    }
}

// physical rows of a thread's logical rows (k is a compile-time constant at every use: the loops are unrolled).  LDS_BACKED: the
// 16-rows-per-thread variant (m > 4096) keeps them in LDS, [k][thread] as 16-bit entries (npad <= 8192) -- in registers they
// were what the kernel spilled
template <int RPT, bool LDS_BACKED> struct PhysRows;
template <int RPT> struct PhysRows<RPT, false> {
    int v[RPT];
    __device__ __forceinline__ PhysRows(unsigned short*) {}
    __device__ __forceinline__ int get(int k) const { return v[k]; }
    __device__ __forceinline__ void set(int k, int x) { v[k] = x; }
};
template <int RPT> struct PhysRows<RPT, true> {
    unsigned short* base;
    __device__ __forceinline__ PhysRows(unsigned short* smem) : base(smem + threadIdx.x) {}
    __device__ __forceinline__ int get(int k) const { return base[k * PT]; }
    __device__ __forceinline__ void set(int k, int x) { base[k * PT] = (unsigned short)x; }
};
template <int RPT, int PWL>
__global__ void __launch_bounds__(PT)
lu_panel_ip_kernel(c128* __restrict__ Hg, c128* __restrict__ Ug, long ld, long strideH, int j0, int m,
                   int* __restrict__ ipiv_g, int* __restrict__ perm_g, int npad, int* __restrict__ info_g)
{
    // tile-major workspace: the 16 panel columns lie inside one 64-column tile, where rows are `ld` = 64 elements apart
    c128* Hm = Hg + (long)blockIdx.x * strideH + lu_tile_off(npad, j0);                    // physical row 0, panel column 0
    c128* Um = Ug + (long)blockIdx.x * strideH + lu_tile_off(npad, j0) + (long)j0 * ld;    // logical row j0, panel column 0
    int* ipiv = ipiv_g + (long)blockIdx.x * npad + j0;
    int* perm = perm_g + (long)blockIdx.x * npad + j0;                     // perm[r]: physical row of panel-local logical row r

    constexpr int LW = NBP - PWL;                 // widest finished part (12 columns)
    __shared__ c128 s_piv[PWL];
    __shared__ c128 s_old[PWL];
    __shared__ double s_val[PT / 64];
    __shared__ int s_idx[PT / 64];
    __shared__ c128 s_L00[LW][LW];
    __shared__ c128 s_U[LW][PWL];
    __shared__ int s_prow[NBP];                   // physical rows of the pivots chosen so far
    __shared__ int s_pphys, s_aphys;
    __shared__ int s_info;
    __shared__ c128 s_rinv;                       // 1 / pivot (1 for an exact zero pivot), by the pivot row's owner
    __shared__ int s_zero;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_info = 0;
#ifdef MAUS_PANEL_CLOCK
    unsigned long long tclk = wall_clock64();
#endif
    c128 R[RPT][PWL];
    __shared__ unsigned short s_phys[(RPT >= 16) ? RPT * PT : 1];
    PhysRows<RPT, (RPT >= 16)> pr(s_phys);
#pragma unroll
    for (int k = 0; k < RPT; ++k) { const int r = tid + k * PT; pr.set(k, (r < m) ? perm[r] : 0); }

    PCLK(0);
#pragma unroll
    for (int sb = 0; sb < NBP / PWL; ++sb) {
        const int c0 = sb * PWL;
        if (sb > 0) {
            // (b') U block of this sub-block's columns above the diagonal block (rows = the pivots chosen so far)
            for (int e = tid; e < c0 * c0; e += PT) { const int j = e / c0, i = e - j * c0; s_L00[j][i] = Hm[(long)s_prow[j] * ld + i]; }
            for (int e = tid; e < c0 * PWL; e += PT) { const int j = e / PWL, c = e - j * PWL; s_U[j][c] = Hm[(long)s_prow[j] * ld + c0 + c]; }
            __syncthreads();
            if (tid < PWL) {
                for (int j = 1; j < c0; ++j) {
                    c128 u = s_U[j][tid];
                    for (int i = 0; i < j; ++i) cfms(u, s_L00[j][i], s_U[i][tid]);
                    s_U[j][tid] = u;
                }
                for (int j = 0; j < c0; ++j) Um[(long)j * ld + c0 + tid] = s_U[j][tid];      // final U entries
            }
            __syncthreads();
        }
        PCLK(1);
        // (c') load this thread's rows of the sub-block, bringing them up to date on the way (pivot rows are done)
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = tid + k * PT;
            if (r < m && r >= c0) {
                const c128* row = Hm + (long)pr.get(k) * ld;
                c128 nw[PWL];
#pragma unroll
                for (int c = 0; c < PWL; ++c) nw[c] = row[c0 + c];
                if (sb > 0) {
#pragma unroll
                    for (int pb = 0; pb < LW / PWL; ++pb) {
                        if (pb < sb) {
                            c128 l[PWL];
#pragma unroll
                            for (int j = 0; j < PWL; ++j) l[j] = row[pb * PWL + j];
#pragma unroll
                            for (int j = 0; j < PWL; ++j)
#pragma unroll
                                for (int c = 0; c < PWL; ++c) cfms(nw[c], l[j], s_U[pb * PWL + j][c]);
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < PWL; ++c) R[k][c] = nw[c];
            }
        }
        PCLK_SYNC(2);
#pragma unroll
        for (int c = 0; c < PWL; ++c) {
            const int a = c0 + c;            // pivot position (panel-local logical row == column index)
            // ---- pivot search: max |re|+|im| over logical rows >= a, first index wins ----
            double best = -1.0; int bidx = INT_MAX;
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int r = tid + k * PT;
                if (r < m && r >= a) {
                    double v = cabs1(R[k][c]);
                    if (v > best) { best = v; bidx = r; }
                }
            }
            wave_argmax(best, bidx);
            if (lane == 0) { s_val[wave] = best; s_idx[wave] = bidx; }
            lds_barrier();
            best = s_val[0]; bidx = s_idx[0];
#pragma unroll
            for (int w = 1; w < PT / 64; ++w) {
                double ov = s_val[w]; int oi = s_idx[w];
                if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
            }
            const int p = (bidx == INT_MAX) ? a : bidx;   // all-NaN column: no interchange (input flagged non-finite)
            // ---- publish pivot row / displaced row of the register sub-block and their physical rows.  Logical row p lives in
            //      thread p % PT as its row p / PT: ONE vector compare finds the owner and the row index is wave-uniform, so
            //      the unrolled `k == kp` tests are scalar branches (round 3 compared every row of every thread with p and with
            //      a, twice per column: ~100 of the ~450 instructions of a column step).  The owner also inverts the pivot --
            //      three fp64 divisions that every thread used to repeat ----
            constexpr int LOG_PT = 9;
            static_assert(PT == 1 << LOG_PT, "owner of a logical row");
            const int kp = p >> LOG_PT, tp = p & (PT - 1);
            if (tid == tp) {
#pragma unroll
                for (int k = 0; k < RPT; ++k) {
                    if (k == kp) {
#pragma unroll
                        for (int cc = 0; cc < PWL; ++cc) s_piv[cc] = R[k][cc];
                        s_pphys = pr.get(k);
                        const c128 pv = R[k][c];
                        const bool zp = (pv.x == 0.0 && pv.y == 0.0);
                        s_rinv = zp ? cmake(1.0, 0.0) : crecip(pv);
                        s_zero = zp ? 1 : 0;
                    }
                }
            }
            if (tid == a && p != a) {                      // logical row a < 16: row 0 of thread a
#pragma unroll
                for (int cc = 0; cc < PWL; ++cc) s_old[cc] = R[0][cc];
                s_aphys = pr.get(0);
            }
            if (tid == 0) ipiv[a] = j0 + p;
            lds_barrier();
            // the interchange: two threads exchange the register rows AND the physical rows they stand for
            if (p != a) {
                if (tid == a) {
#pragma unroll
                    for (int cc = 0; cc < PWL; ++cc) R[0][cc] = s_piv[cc];
                    pr.set(0, s_pphys);
                }
                if (tid == tp) {
#pragma unroll
                    for (int k = 0; k < RPT; ++k) {
                        if (k == kp) {
#pragma unroll
                            for (int cc = 0; cc < PWL; ++cc) R[k][cc] = s_old[cc];
                            pr.set(k, s_aphys);
                        }
                    }
                }
            }
            if (tid == 0) s_prow[a] = s_pphys;
            // row a of U inside this sub-block (columns >= a); the columns of later sub-blocks follow in their (b')
            if (tid < PWL && tid >= c) Um[(long)a * ld + c0 + tid] = s_piv[tid];
            if (s_zero && tid == 0 && s_info == 0) s_info = j0 + a + 1;   // LAPACK info (1-based)
            const c128 rinv = s_rinv;
            c128 prow[PWL];
#pragma unroll
            for (int cc = 0; cc < PWL; ++cc) prow[cc] = s_piv[cc];
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int r = tid + k * PT;
                if (r < m && r > a) {
                    c128 l = cmul(R[k][c], rinv);
                    R[k][c] = l;
#pragma unroll
                    for (int cc = c + 1; cc < PWL; ++cc) cfms(R[k][cc], l, prow[cc]);
                }
            }
            // no barrier here: the next column rewrites s_val behind this column's second barrier (everybody has read it)
            // and s_piv / s_pphys behind its own first barrier (everybody has finished this update)
        }
        PCLK(3);
        // store the factored sub-block (L below the pivots; the pivot rows' own entries are only read back by (b'))
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = tid + k * PT;
            if (r < m && r >= c0) {
                c128* row = Hm + (long)pr.get(k) * ld;
#pragma unroll
                for (int c = 0; c < PWL; ++c) row[c0 + c] = R[k][c];
            }
        }
        __syncthreads();       // the next sub-block reads these columns from memory
        PCLK(4);
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) { const int r = tid + k * PT; if (r < m) perm[r] = pr.get(k); }
    if (tid == 0 && s_info != 0 && info_g[blockIdx.x] == 0) info_g[blockIdx.x] = s_info;
    // the diagonal block's inverse: its rows are the 16 pivot rows, stored sub-block by sub-block (the barrier behind the last
    // store has made them visible to the whole workgroup)
    __shared__ c128 s_diag[16][17];
    if (tid < 256) { const int r = tid >> 4, c = tid & 15; s_diag[r][c] = Hm[(long)s_prow[r] * ld + c]; }
    __syncthreads();
    panel_store_diaginv(s_diag, Um, ld, tid);
}

#if MAUS_NBP == 16
// ---------------------------------------------------------------------------------------
// Base panel for m <= 1024 (round 3): one workgroup per matrix keeps its WHOLE 16-column slice in registers -- one or two
// rows of 16 columns per thread -- so the panel is read once and written once (32 column passes; the left-looking kernel
// above re-reads its finished columns: 40-56 passes) and factored right-looking: per pivot column one search, one
// exchange of the pivot row through LDS, one rank-1 update of the columns to its right.  Same pivot rule (izamax over
// logical rows, first index wins), same implicit interchange (register rows and perm entries swap) as the other panels.
// Every panel of a 1024 x 1024 factorisation (BASELINE configs[1]) and the last quarter of the panels at n = 4096.
// NT threads own RPT x NT rows: the workgroup shrinks with the panel (128 / 256 / 512 threads for m <= 256 / 512 / 1024), so
// that several matrices share a CU -- a 512-thread workgroup at 177 VGPRs has a CU to itself, and a batch of more matrices
// than CUs then runs its 16 latency-bound pivot steps in two rounds.
// ---------------------------------------------------------------------------------------
template <int RPT, int NT>
__global__ void __launch_bounds__(NT)
lu_panel_rs_kernel(c128* __restrict__ Hg, c128* __restrict__ Ug, long ld, long strideH, int j0, int m,
                   int* __restrict__ ipiv_g, int* __restrict__ perm_g, int npad, int* __restrict__ info_g)
{
    const int g = blockIdx.x;
    c128* Hm = Hg + (long)g * strideH + lu_tile_off(npad, j0);                             // tile-major: see lu_panel_ip_kernel
    c128* Um = Ug + (long)g * strideH + lu_tile_off(npad, j0) + (long)j0 * ld;
    int* ipiv = ipiv_g + (long)g * npad + j0;
    int* perm = perm_g + (long)g * npad + j0;                     // perm[r]: physical row of panel-local logical row r

    __shared__ double s_val[NT / 64];
    __shared__ int s_idx[NT / 64];
    __shared__ c128 s_row[NBP];          // the pivot row
    __shared__ c128 s_arow[NBP];         // logical row a (displaced by the interchange)
    __shared__ int s_phys[2];            // physical rows of the pivot row / of logical row a
    __shared__ c128 s_rinv;              // 1 / pivot (1 for an exact zero pivot), by the pivot row's owner
    __shared__ int s_zero;
    static_assert((NT & (NT - 1)) == 0, "owner of a logical row");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    c128 R[RPT][NBP];
    int pr[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int r = tid + k * NT;
        pr[k] = (r < m) ? perm[r] : 0;
        if (r < m) {
            const c128* row = Hm + (long)pr[k] * ld;
#pragma unroll
            for (int c = 0; c < NBP; ++c) R[k][c] = row[c];
        }
    }
    int my_info = 0;
#ifdef MAUS_PANEL_CLOCK
    unsigned long long tclk = wall_clock64();
#endif
    PCLK_SYNC(0);
    auto step = [&](auto AC) {
        constexpr int a = decltype(AC)::value;
        // ---- pivot search: max |re|+|im| over logical rows >= a, first index wins ----
        double best = -1.0; int bidx = INT_MAX;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = tid + k * NT;
            if (r < m && r >= a) {
                double v = cabs1(R[k][a]);
                if (v > best) { best = v; bidx = r; }
            }
        }
        wave_argmax(best, bidx);
        if (lane == 0) { s_val[wave] = best; s_idx[wave] = bidx; }
        lds_barrier();
        PCLK(5);
        best = s_val[0]; bidx = s_idx[0];
#pragma unroll
        for (int q = 1; q < NT / 64; ++q) {
            double ov = s_val[q]; int oi = s_idx[q];
            if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
        }
        const int p = (bidx == INT_MAX) ? a : bidx;   // all-NaN column: no interchange (input flagged non-finite)
        // ---- publish the pivot row and the row it displaces: by their owners (thread p % NT, row p / NT: see
        //      lu_panel_ip_kernel); the pivot row's owner inverts the pivot for everybody ----
        const int kp = p / NT, tp = p % NT;                  // NT is a power of two
        if (tid == tp) {
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                if (k == kp) {
#pragma unroll
                    for (int c = 0; c < NBP; ++c) s_row[c] = R[k][c];
                    s_phys[0] = pr[k];
                    const c128 pv = R[k][a];
                    const bool zp = (pv.x == 0.0 && pv.y == 0.0);
                    s_rinv = zp ? cmake(1.0, 0.0) : crecip(pv);
                    s_zero = zp ? 1 : 0;
                }
            }
        }
        if (tid == a && p != a) {                            // logical row a < 16 <= NT: row 0 of thread a
#pragma unroll
            for (int c = 0; c < NBP; ++c) s_arow[c] = R[0][c];
            s_phys[1] = pr[0];
        }
        lds_barrier();
        PCLK(6);
        // ---- the interchange: the owners of logical rows a and p exchange register rows and physical rows ----
        if (p != a) {
            if (tid == a) {
#pragma unroll
                for (int c = 0; c < NBP; ++c) R[0][c] = s_row[c];
                pr[0] = s_phys[0];
            }
            if (tid == tp) {
#pragma unroll
                for (int k = 0; k < RPT; ++k) {
                    if (k == kp) {
#pragma unroll
                        for (int c = 0; c < NBP; ++c) R[k][c] = s_arow[c];
                        pr[k] = s_phys[1];
                    }
                }
            }
        }
        if (s_zero && my_info == 0) my_info = j0 + a + 1;                // LAPACK info (1-based)
        if (tid == 0) ipiv[a] = j0 + p;
        if (tid < NBP && tid >= a) Um[(long)a * ld + tid] = s_row[tid];   // row a of U inside the panel
        const c128 rinv = s_rinv;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = tid + k * NT;
            if (r < m && r > a) {
                const c128 l = cmul(R[k][a], rinv);
                R[k][a] = l;
#pragma unroll
                for (int c = 0; c < NBP; ++c) if (c > a) cfms(R[k][c], l, s_row[c]);
            }
        }
        PCLK(7);
        // two barriers per column are enough: the next column rewrites s_val behind this column's second barrier
        // (everybody has read it) and s_row behind its own first barrier (everybody has finished this update)
    };
#define RS_STEP(A) step(std::integral_constant<int, A>{})
    RS_STEP(0); RS_STEP(1); RS_STEP(2); RS_STEP(3); RS_STEP(4); RS_STEP(5); RS_STEP(6); RS_STEP(7);
    RS_STEP(8); RS_STEP(9); RS_STEP(10); RS_STEP(11); RS_STEP(12); RS_STEP(13); RS_STEP(14); RS_STEP(15);
#undef RS_STEP
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int r = tid + k * NT;
        if (r < m) {
            c128* row = Hm + (long)pr[k] * ld;
#pragma unroll
            for (int c = 0; c < NBP; ++c) row[c] = R[k][c];
            perm[r] = pr[k];
        }
    }
    if (tid == 0 && my_info != 0 && info_g[g] == 0) info_g[g] = my_info;
    // the diagonal block's inverse: logical row i < 16 is row 0 of thread i
    __shared__ c128 s_diag[16][17];
    if (tid < 16) {
#pragma unroll
        for (int c = 0; c < NBP; ++c) s_diag[tid][c] = R[0][c];
    }
    __syncthreads();
    panel_store_diaginv(s_diag, Um, ld, tid);
    PCLK_SYNC(4);
}

// ---------------------------------------------------------------------------------------
// Base panel spread over W workgroups per matrix (small batches: with one workgroup per matrix a batch of 32 occupies 32
// of the 256 CUs and the panel phase is bound by one CU's memory pipeline per matrix).  Workgroup w owns the logical rows
// [w*RPT*PT, (w+1)*RPT*PT) of the panel and keeps its whole 16-column slice in registers (read once, written once); the
// pivot search and the exchange of the pivot row go through a small per-matrix area in global memory, ONE hand-off per column
// (round 4; rounds 2-3 needed three dependent trips through L2: publish + drain, arrive on a counter and wait, read the
// candidates):
//   per column a:  every workgroup publishes its best row -- |re|+|im|, logical index, physical row, the 16 entries -- and, if
//                  it owns logical row a, that row too, as 8-byte GRANULES {tag = (panel, column), 32 bits of payload}, each written by
//                  one agent-scope atomic store; every workgroup polls all granules of the column (one per thread) until each
//                  carries the column's tag.  A granule validates itself -- data and tag arrive in one store -- so there is no
//                  flag, no counter and nothing to drain (MI355X_MICROARCH.md: data-tagged granules, handoff-1to1).  The winner
//                  is then picked out of LDS by the izamax rule (max value, lowest logical index).
// Granule buffers alternate with the column parity: nobody can be two columns ahead, because publishing column a + 2 takes the
// candidates of column a + 1 from everybody, and those are published only after their owners have read column a.
// Every wait is bounded: after ~2 s without progress a thread raises the abort word, everybody leaves, and the
// matrix reports info = INT_MIN (internal error) instead of hanging the device.  The launcher only uses this kernel
// when all G*W workgroups are co-resident by construction (one launch in flight, G*W <= number of CUs).
// ---------------------------------------------------------------------------------------
constexpr int MW_MAXW = 8;
constexpr int MW_GRAN = 4 + 4 * NBP;    // payload words of one record: value lo / hi, physical row, logical row, 16 complex entries
struct MwSync {                         // one per matrix; zeroed at the start of every factorisation (tags are unique inside one)
    unsigned long long abort_;          // non-zero: a wait timed out
    unsigned long long gran[2][MW_MAXW + 1][MW_GRAN];    // [parity][workgroup w; slot MW_MAXW = logical row a][word]: tag << 32 | payload
};

__device__ __forceinline__ void mw_store(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long mw_load(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int RPT>
__global__ void __launch_bounds__(PT)
lu_panel_mw_kernel(c128* __restrict__ Hg, c128* __restrict__ Ug, long ld, long strideH, int j0, int m, int W,
                   int* __restrict__ ipiv_g, int* __restrict__ perm_g, int npad, int* __restrict__ info_g, MwSync* __restrict__ sync_g,
                   unsigned long long timeout_ticks, int force_abort, unsigned tag_base)
{
    // blockIdx.x = matrix, blockIdx.y = row chunk: workgroups are dealt round-robin over the 8 XCDs by linear id, so with a
    // batch that is a multiple of 8 the W workgroups of one matrix share an XCD -- and its L2 -- (speed only)
    const int w = blockIdx.y, g = blockIdx.x;
    c128* Hm = Hg + (long)g * strideH + lu_tile_off(npad, j0);                             // tile-major: see lu_panel_ip_kernel
    c128* Um = Ug + (long)g * strideH + lu_tile_off(npad, j0) + (long)j0 * ld;
    int* ipiv = ipiv_g + (long)g * npad + j0;
    int* perm = perm_g + (long)g * npad + j0;                     // perm[r]: physical row of panel-local logical row r
    MwSync* sy = sync_g + g;

    __shared__ double s_val[PT / 64];
    __shared__ int s_idx[PT / 64];
    __shared__ c128 s_row[NBP];          // this workgroup's candidate row, then the winner's row
    __shared__ c128 s_arow[NBP];         // logical row a (published by its owner), then the displaced row
    __shared__ int s_phys[2];            // [0] candidate / winner physical row, [1] physical row of logical row a
    __shared__ unsigned long long s_meta[MW_MAXW][2];
    __shared__ double s_all[MW_MAXW][2 * NBP];       // every workgroup's candidate row
    __shared__ int s_abort;
    __shared__ c128 s_rinv;              // 1 / pivot (1 for an exact zero pivot)
    __shared__ int s_zero;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rbase = w * RPT * PT;
    if (tid == 0) s_abort = 0;
    c128 R[RPT][NBP];
    int pr[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int r = rbase + tid + k * PT;
        pr[k] = (r < m) ? perm[r] : 0;
        if (r < m) {
            const c128* row = Hm + (long)pr[k] * ld;
#pragma unroll
            for (int c = 0; c < NBP; ++c) R[k][c] = row[c];
        }
    }
#ifdef MAUS_PANEL_CLOCK
    unsigned long long tclk = wall_clock64();
#endif
    int my_info = 0;
    bool aborted = false;         // workgroup-uniform: a rendezvous timed out (here or in a sibling workgroup)
    __syncthreads();

    // (the barriers inside a column step order LDS traffic only -- lds_barrier -- : what travels through global memory is drained by
    // its writer's own s_waitcnt and reaches LDS through a data dependence)
    // one instantiation per column (R is indexed by the column, so `a` must be a compile-time constant; a 16-fold
    // `#pragma unroll` of this body exceeds the unroller's size limit); after an abort the remaining steps are skipped
    PCLK(8);
    auto step = [&](auto AC) {
      constexpr int a = decltype(AC)::value;
      if (!aborted) {
        const int par = a & 1;
        // ---- local candidate: max |re|+|im| over own logical rows >= a, first index wins ----
        double best = -1.0; int bidx = INT_MAX;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = rbase + tid + k * PT;
            if (r < m && r >= a) {
                double v = cabs1(R[k][a]);
                if (v > best) { best = v; bidx = r; }
            }
        }
        wave_argmax(best, bidx);
        if (lane == 0) { s_val[wave] = best; s_idx[wave] = bidx; }
        lds_barrier();
        best = s_val[0]; bidx = s_idx[0];
#pragma unroll
        for (int q = 1; q < PT / 64; ++q) {
            double ov = s_val[q]; int oi = s_idx[q];
            if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
        }
        const bool own_a = (w == 0);                 // logical rows 0..15 belong to the first workgroup
        // (owners: local row q = r - rbase lives in thread q % PT as its row q / PT, see lu_panel_ip_kernel)
        if (bidx != INT_MAX && tid == ((bidx - rbase) & (PT - 1))) {
            const int kb = (bidx - rbase) >> 9;
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                if (k == kb) {
#pragma unroll
                    for (int c = 0; c < NBP; ++c) s_row[c] = R[k][c];
                    s_phys[0] = pr[k];
                }
            }
        }
        if (own_a && tid == a) {
#pragma unroll
            for (int c = 0; c < NBP; ++c) s_arow[c] = R[0][c];
            s_phys[1] = pr[0];
        }
        lds_barrier();
        PCLK(9);
        // ---- publish: one granule per thread, no drain, no flag ----
        const unsigned tag = tag_base | (unsigned)(a + 1);             // unique within a factorisation: (panel + 1) << 5 | column + 1
        const unsigned long long tagw = (unsigned long long)tag << 32;
        // test hook (MAUS_PANEL_MW_FORCE_ABORT): the last workgroup of matrix 0 publishes nothing for column 3, so its
        // siblings -- and then itself, one column later -- run into the time-out exactly as if it had never become resident
        const bool mute = force_abort && g == 0 && w == W - 1 && a == 3;
        if (tid < MW_GRAN && !mute) {
            unsigned pay;
            if (tid == 0) pay = (unsigned)(__double_as_longlong(best) & 0xffffffffll);
            else if (tid == 1) pay = (unsigned)((unsigned long long)__double_as_longlong(best) >> 32);
            else if (tid == 2) pay = (bidx == INT_MAX) ? 0u : (unsigned)s_phys[0];
            else if (tid == 3) pay = (unsigned)bidx;
            else pay = ((const unsigned*)s_row)[tid - 4];
            mw_store(&sy->gran[par][w][tid], tagw | pay);
        }
        if (own_a && tid >= 64 && tid < 64 + MW_GRAN && !mute) {
            const int i = tid - 64;
            const unsigned pay = (i == 2) ? (unsigned)s_phys[1] : (i >= 4 ? ((const unsigned*)s_arow)[i - 4] : 0u);
            mw_store(&sy->gran[par][MW_MAXW][i], tagw | pay);
        }
        lds_barrier();                  // s_row / s_arow / s_phys have been read: the gather below overwrites them
        PCLK(10);
        // ---- gather: every granule of the column, one per thread, polled until it carries the column's tag.  (Polling the W
        //      metas first and then only the winner's row -- a quarter of the pollers -- was slower, 22.0 vs 19.9 ms of panel
        //      time per 32-solve call: the second, dependent trip costs more than the extra pollers.) ----
        {
            const int total = (W + 1) * MW_GRAN;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
            for (int e = tid; e < total; e += PT) {
                const int q = e / MW_GRAN, i = e - q * MW_GRAN;
                const unsigned long long* gp_ = &sy->gran[par][q < W ? q : MW_MAXW][i];
                unsigned long long v;
                int spins = 0;
                bool got = true;
                while ((unsigned)(v = mw_load(gp_), v >> 32) != tag) {
                    if ((++spins & 255) == 0) {
                        if (mw_load(&sy->abort_)) { got = false; break; }
                        if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) { mw_store(&sy->abort_, 1ull); got = false; break; }
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!got) { s_abort = 1; break; }
                const unsigned pay = (unsigned)(v & 0xffffffffull);
                if (q < W) {
                    if (i < 4) ((unsigned*)s_meta)[q * 4 + i] = pay;            // [value lo, value hi, physical row, logical row]
                    else ((unsigned*)s_all)[q * (4 * NBP) + (i - 4)] = pay;
                } else if (i == 2) s_phys[1] = (int)pay;
                else if (i >= 4) ((unsigned*)s_arow)[i - 4] = pay;
            }
        }
        lds_barrier();
        PCLK(11);
        aborted = (s_abort != 0);
      }
      if (!aborted) {
        const bool own_a = (w == 0);
        PCLK(12);
        double gv = -1.0; int gp = INT_MAX, gw = 0, gphys = 0;
        for (int q = 0; q < W; ++q) {
            const double v = __longlong_as_double((long long)s_meta[q][0]);
            const int idx = (int)(s_meta[q][1] >> 32);
            if (idx != INT_MAX && (v > gv || (v == gv && idx < gp))) { gv = v; gp = idx; gw = q; gphys = (int)(unsigned)(s_meta[q][1] & 0xffffffffull); }
        }
        const bool none = (gp == INT_MAX);              // all-NaN column: no interchange (input flagged non-finite)
        const int p = none ? a : gp;
        if (tid < 2 * NBP) ((double*)s_row)[tid] = none ? ((const double*)s_arow)[tid] : s_all[gw][tid];
        if (tid == 0) s_phys[0] = none ? s_phys[1] : gphys;
        if (tid == 64) {                                 // the pivot's inverse, once (three fp64 divisions every thread used to repeat)
            const double* src = none ? (const double*)s_arow : s_all[gw];
            const c128 pv = cmake(src[2 * a], src[2 * a + 1]);
            const bool zp = (pv.x == 0.0 && pv.y == 0.0);
            s_rinv = zp ? cmake(1.0, 0.0) : crecip(pv);
            s_zero = zp ? 1 : 0;
        }
        lds_barrier();
        PCLK(13);
        // ---- the interchange: the owners of logical rows a and p exchange register rows and physical rows ----
        if (p != a) {
            if (own_a && tid == a) {
#pragma unroll
                for (int c = 0; c < NBP; ++c) R[0][c] = s_row[c];
                pr[0] = s_phys[0];
            }
            const int q = p - rbase;                     // the winner lives in this workgroup iff 0 <= q < RPT * PT
            if (q >= 0 && q < RPT * PT && tid == (q & (PT - 1))) {
                const int kq = q >> 9;
#pragma unroll
                for (int k = 0; k < RPT; ++k) {
                    if (k == kq) {
#pragma unroll
                        for (int c = 0; c < NBP; ++c) R[k][c] = s_arow[c];
                        pr[k] = s_phys[1];
                    }
                }
            }
        }
        if (s_zero && my_info == 0) my_info = j0 + a + 1;                // LAPACK info (1-based)
        if (own_a) {
            if (tid == 0) ipiv[a] = j0 + p;
            if (tid < NBP && tid >= a) Um[(long)a * ld + tid] = s_row[tid];       // row a of U inside the panel
        }
        const c128 rinv = s_rinv;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = rbase + tid + k * PT;
            if (r < m && r > a) {
                const c128 l = cmul(R[k][a], rinv);
                R[k][a] = l;
#pragma unroll
                for (int c = 0; c < NBP; ++c) if (c > a) cfms(R[k][c], l, s_row[c]);
            }
        }
        lds_barrier();
        PCLK(14);
      }
    };
    static_assert(NBP == 16, "lu_panel_mw_kernel: 16 column steps");
#define MW_STEP(A) step(std::integral_constant<int, A>{})
    MW_STEP(0); MW_STEP(1); MW_STEP(2); MW_STEP(3); MW_STEP(4); MW_STEP(5); MW_STEP(6); MW_STEP(7);
    MW_STEP(8); MW_STEP(9); MW_STEP(10); MW_STEP(11); MW_STEP(12); MW_STEP(13); MW_STEP(14); MW_STEP(15);
#undef MW_STEP
    if (aborted) { if (tid == 0 && w == 0) info_g[g] = INT_MIN; return; }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int r = rbase + tid + k * PT;
        if (r < m) {
            c128* row = Hm + (long)pr[k] * ld;
#pragma unroll
            for (int c = 0; c < NBP; ++c) row[c] = R[k][c];
            perm[r] = pr[k];
        }
    }
    if (tid == 0 && w == 0 && my_info != 0 && info_g[g] == 0) info_g[g] = my_info;
    if (w == 0) {                        // the diagonal block's inverse: logical row i < 16 is row 0 of thread i of the first workgroup
        __shared__ c128 s_diag[16][17];
        if (tid < 16) {
#pragma unroll
            for (int c = 0; c < NBP; ++c) s_diag[tid][c] = R[0][c];
        }
        __syncthreads();
        panel_store_diaginv(s_diag, Um, ld, tid);
    }
    PCLK(15);
}

#endif  // MAUS_NBP == 16

// =======================================================================================
// Triangular solves of the block rows,  U[j : j+k, cols] = L11^-1 * H[perm[j : j+k), cols],  k <= 128, on the matrix pipe
// (round 4).  Round 3 ran them as a recursion of 32-row substitution kernels (one thread per column) and K = 32 / 64 zgemm
// updates in between: 6.5 passes over every block row for its levels below 128, 15 dependent launches per 256-row solve.
// Now one kernel per solve of up to 128 rows reads the block row ONCE and writes it ONCE:
//   * one wave owns a strip of 16 columns and keeps all k x 16 entries of it in registers, in the accumulator layout of
//     v_mfma_f64_16x16x4 (component r of block bi = row 16 bi + 4 r + lane/16, column lane%16).  That layout IS the B-operand
//     layout of the same instruction (k-step s takes rows 4 s + lane/16), so a finished block X_bj feeds the products of the
//     blocks below it without leaving the registers;
//   * block forward substitution over 16-row blocks:  X_bi = Dinv_bi (B_bi - sum_{bj<bi} L[bi][bj] X_bj), every 16 x 16 complex
//     block product as 3 real MFMA products (3M, the zgemm's form and rounding order); the L blocks are A operands read
//     straight from the factored panel columns through perm (L2-resident: every strip of a matrix reads the same blocks);
//   * Dinv_bi = inverse of the unit lower triangular 16 x 16 diagonal block (LAPACK's own large-n zgetrf reaches its U12
//     through ztrsm, which libraries implement with inverted diagonal blocks as well).  |l| <= 1 under partial pivoting.  Its
//     strictly lower part is computed once per block at the end of the panel kernel that factored it (panel_store_diaginv)
//     and kept where nothing else lives: below the
//     diagonal of the same block of the logical-order U array.
// All B tiles of a strip are requested before the first product (32 loads per lane in flight).
// =======================================================================================
template <int NB>
__global__ void __launch_bounds__(256, 2)
trsm_mfma_kernel(const c128* __restrict__ Hg, c128* __restrict__ Ug, long strideH, const int* __restrict__ perm_g,
                 int npad, int j, int c_lo, int c_hi)
{
    const int g = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col0 = c_lo + (blockIdx.x * 4 + wave) * 16;                   // this wave's strip (never straddles a 64-column tile)
    if (col0 >= c_hi) return;                                               // wave-uniform; the kernel has no barrier
    const c128* H = Hg + (long)g * strideH;
    c128* U = Ug + (long)g * strideH;
    const int* perm = perm_g + (long)g * npad + j;
    const int q = lane >> 4, i16 = lane & 15;
    // addresses = wave-uniform 64-bit base + 32-bit lane offset (a physical row is at most 8192 * 1 KB into a tile): one VGPR per
    // address instead of two keeps the 4 NB tile loads that are in flight together inside the register budget
    const c128* Hs = H + lu_tile_off(npad, col0);                            // this strip's tile column, physical row 0
    c128* Us = U + lu_tile_off(npad, col0) + (long)j * LU_TW;                // this strip's tile column, logical row j
    // the row lists first, then every B tile of the strip: 4 NB loads per lane in flight before the first product
    // (the row offsets of the L fragments wait in LDS, one private slot per lane and block row: eight registers fewer)
    __shared__ unsigned s_lrow[4][NB][64];
    // (plain doubles, not d4 vectors: a B operand of the MFMA is any register pair, so the tiles stay where their loads put them)
    double xr[NB][4], xi[NB][4];
    {
        unsigned prow[NB][4];
#pragma unroll
        for (int bi = 0; bi < NB; ++bi) {
#pragma unroll
            for (int r = 0; r < 4; ++r) prow[bi][r] = (unsigned)perm[16 * bi + 4 * r + q] * LU_TW + i16;
            s_lrow[wave][bi][lane] = (unsigned)perm[16 * bi + i16] * LU_TW + q;
        }
#pragma unroll
        for (int bi = 0; bi < NB; ++bi) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const c128 v = Hs[prow[bi][r]];
                xr[bi][r] = v.x; xi[bi][r] = v.y;
            }
        }
    }
    // A-operand fragments: lane holds block element [row i16][column 4 s + q], s = 0..3
    auto load_l = [&](int bi, int bj, c128 (&f)[4]) {
        const c128* p = H + lu_tile_off(npad, j + 16 * bj);                  // uniform
        const unsigned lr = s_lrow[wave][bi][lane];
#pragma unroll
        for (int s = 0; s < 4; ++s) f[s] = p[lr + 4 * s];
    };
    const unsigned drow = (unsigned)i16 * LU_TW + q;
    auto load_d = [&](int bi, c128 (&f)[4]) {          // strictly lower part from memory, unit diagonal, zeros above
        const c128* p = U + lu_tile_off(npad, j + 16 * bi) + (long)(j + 16 * bi) * LU_TW;     // uniform
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = 4 * s + q;
            const c128 v = p[drow + 4 * s];
            f[s] = cmake(k < i16 ? v.x : (k == i16 ? 1.0 : 0.0), k < i16 ? v.y : 0.0);
        }
    };
    // One flat sequence of block products -- for bi = 0, 1, ..: L[bi][0], .., L[bi][bi-1], then Dinv_bi -- with the A fragments of
    // the NEXT product requested before the current one runs, across the bi boundaries too, and the stores of a finished block
    // issued behind that request: vmcnt retires in order, so a load requested after the stores would wait for them.
    auto load_blk = [&](int bi, int bj, c128 (&f)[4]) { if (bj < bi) load_l(bi, bj, f); else load_d(bi, f); };
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    c128 cur[4];
    load_d(0, cur);
#pragma unroll
    for (int bi = 0; bi < NB; ++bi) {
        {
            d4 s1 = zero, s2 = zero, s3 = zero;
#pragma unroll
            for (int bj = 0; bj <= bi; ++bj) {
                c128 nx[4];
                const bool last = (bj == bi);
                const int nbi = last ? bi + 1 : bi, nbj = last ? 0 : bj + 1;
                const bool have_next = nbi < NB;
                if (have_next) load_blk(nbi, nbj, nx);
                if (!last) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        s1 = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[s].x, xr[bj][s], s1, 0, 0, 0);
                        s2 = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[s].y, xi[bj][s], s2, 0, 0, 0);
                        s3 = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[s].x + cur[s].y, xr[bj][s] + xi[bj][s], s3, 0, 0, 0);
                    }
                } else {
                    // the zgemm's 3M epilogue (re = P1 - P2, im = (P3 - P1) - P2), subtracted from the B tile
                    double tr[4], ti[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        tr[s] = (bi > 0) ? xr[bi][s] - (s1[s] - s2[s]) : xr[bi][s];
                        ti[s] = (bi > 0) ? xi[bi][s] - ((s3[s] - s1[s]) - s2[s]) : xi[bi][s];
                    }
                    d4 p1 = zero, p2 = zero, p3 = zero;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[s].x, tr[s], p1, 0, 0, 0);
                        p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[s].y, ti[s], p2, 0, 0, 0);
                        p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[s].x + cur[s].y, tr[s] + ti[s], p3, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        xr[bi][r] = p1[r] - p2[r];
                        xi[bi][r] = (p3[r] - p1[r]) - p2[r];
                        Us[(unsigned)((16 * bi + 4 * r + q) * LU_TW + i16)] = cmake(xr[bi][r], xi[bi][r]);
                    }
                }
                if (have_next) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) cur[s] = nx[s];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Back substitution U x = y (y = augmented column npad).  One workgroup per matrix,
// 32-row blocks from the bottom: dot products of the U row tails against x (LDS), then a
// 32x32 triangle solved by one wave.  Writes x[0..n) to W[slot]; flags bit1 <- non-finite x.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512)
backsolve_kernel(c128* __restrict__ Hg, long ld, long strideH, int n, int npad,
                 c128* __restrict__ Wg, long ldw, const int* __restrict__ slots, c128* __restrict__ xout_dense,
                 int* __restrict__ flags, int row_lo, int row_hi)
{
    // rows [row_lo, row_hi): the columns from row_hi on have already been taken out of y (blocked form, maus_lu_backsolve), and
    // x then goes back into the augmented column, where the update of the rows above reads it; the whole matrix in one
    // launch is row_lo = 0, row_hi = npad
    const bool whole = (row_lo == 0 && row_hi == npad);
    extern __shared__ c128 sx[];          // npad entries of x, then a 32x33 diagonal block, then 2 x 32 partial sums
    c128* sD = sx + npad;
    c128* sP = sD + BSB * (BSB + 1);           // 2 x 32 partial sums of the block's rows
    const int g = blockIdx.x;
    c128* H = Hg + (long)g * strideH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bool bad = false;
    for (int i0 = row_hi - BSB; i0 >= row_lo; i0 -= BSB) {
        const int jt = i0 + BSB;          // tail starts here
        // rhs_i = y_i - U[i, jt:] . x[jt:] for the block's 32 rows.  Round 4 (later): the 8 waves are four row groups of 8 rows
        // times two column groups (tiles t = t_lo + cg, + 2, ..), every lane with the 8 rows of FOUR tiles in flight at once
        // (32 loads), and the partial sums meet in LDS.  Rounds 1-3 gave each of 16 waves two rows and the whole tail: 16
        // dependent iterations of 8 loads at a tail of 2048 columns where this makes 4 of 32, and the chain of 128 blocks is
        // what a call of few matrices waits for (3.9 ms at any batch below ~100; at 181 the kernel is HBM-bound either way).
        // (the diagonal block is requested first: its latency passes behind the tail products)
        for (int e = tid; e < BSB * BSB; e += blockDim.x) {
            int r = e / BSB, c = e % BSB;
            sD[r * (BSB + 1) + c] = H[lu_tile_off(npad, i0 + c) + (long)(i0 + r) * ld];
        }
        {
            static_assert(BSB == 32, "four row groups of 8");
            const int rg = wave & 3, cg = wave >> 2;
            const c128* rbase = H + (long)(i0 + 8 * rg) * ld;             // tile-major (luws.h): + tile offset of the column
            c128 acc[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] = cmake(0.0, 0.0);
            const int t_lo = jt >> 6, t_hi = (row_hi + 63) >> 6;
            for (int t = t_lo + cg; t < t_hi; t += 8) {
                c128 u[4][8], xv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int tq = t + 2 * q, j = (tq << 6) + lane;
                    const bool ok = tq < t_hi && j >= jt && j < row_hi;
                    xv[q] = ok ? sx[j] : cmake(0.0, 0.0);
                    const long off = ((long)tq * npad << 6) + lane;
#pragma unroll
                    for (int r = 0; r < 8; ++r) u[q][r] = ok ? rbase[(long)r * ld + off] : cmake(0.0, 0.0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) cfma(acc[r], u[q][r], xv[q]);
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const double sr = wave_sum_dpp(acc[r].x), si = wave_sum_dpp(acc[r].y);
                if (lane == 0) sP[cg * BSB + 8 * rg + r] = cmake(sr, si);
            }
        }
        __syncthreads();
        if (wave == 0) {
            c128 rv = cmake(0.0, 0.0), dinv = cmake(0.0, 0.0);
            if (lane < BSB) {
                const c128 sum = cadd(sP[lane], sP[BSB + lane]);
                const c128 y = H[(long)(i0 + lane) * ld + lu_tile_off(npad, npad)];
                rv = cmake(y.x - sum.x, y.y - sum.y);
                dinv = crecip(sD[lane * (BSB + 1) + lane]);      // all 32 diagonal entries inverted at once: the substitution
            }                                                     // below multiplies (rounds 1-3 divided, one division chain per row)
            // 32 dependent steps: the lane index of the broadcast is a compile-time constant (v_readlane, not ds_bpermute)
#pragma unroll
            for (int j = BSB - 1; j >= 0; --j) {
                c128 xj = cmul(rv, dinv);                         // meaningful on lane j only
                xj.x = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xj.x), j), __builtin_amdgcn_readlane(__double2loint(xj.x), j));
                xj.y = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xj.y), j), __builtin_amdgcn_readlane(__double2loint(xj.y), j));
                if (lane == j) { sx[i0 + j] = xj; if (!whole) H[(long)(i0 + j) * ld + lu_tile_off(npad, npad)] = xj; }
                if (lane < j) cfms(rv, sD[lane * (BSB + 1) + j], xj);
            }
        }
        __syncthreads();
    }
    if (!whole) return;
    c128* out = (Wg != nullptr) ? Wg + (long)slots[g] * ldw : xout_dense + (long)g * n;
    for (int i = tid; i < n; i += blockDim.x) { c128 v = sx[i]; bad |= !cfinite(v); out[i] = v; }
    if (__any(bad) && lane == 0) atomicOr(&flags[g], 2);
}

// x out of the augmented column (blocked back substitution): W[slot] or the dense output, flags bit1 <- non-finite x
__global__ void __launch_bounds__(256)
backsolve_out_kernel(const c128* __restrict__ Hg, long ld, long strideH, int n, int npad, c128* __restrict__ Wg, long ldw,
                     const int* __restrict__ slots, c128* __restrict__ xout_dense, int* __restrict__ flags)
{
    const int g = blockIdx.y;
    const c128* col = Hg + (long)g * strideH + lu_tile_off(npad, npad);
    c128* out = (Wg != nullptr) ? Wg + (long)slots[g] * ldw : xout_dense + (long)g * n;
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool bad = false;
    if (i < n) { const c128 v = col[(long)i * ld]; bad = !cfinite(v); out[i] = v; }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&flags[g], 2);
}

}  // namespace

// =======================================================================================
// host-side drivers (device pointers, all batched over G matrices on stream st)
// =======================================================================================

static inline void prof(const LuWs& w, int klass, int phase, double flops = 0, double bytes = 0) { if (w.tick) w.tick(w.ud, klass, phase, flops, bytes); }

static void lu_gemm(const LuWs& w, int r0, int r1, int c0, int c1, int k0, int k1) {
    // H[r0:r1, c0:c1] -= H[r0:r1, k0:k1] * H[k0:k1, c0:c1]   (implicit pivoting: rows r0:r1 through perm, the
    // k0:k1 rows of the right factor are finished U rows and come from the logical-order array)
    int M = r1 - r0, N = c1 - c0, K = k1 - k0;
    if (M <= 0 || N <= 0 || K <= 0) return;
    const int kc = (K >= 256) ? KC_GEMM : (K >= 128) ? KC_GEMM_K128 : (K >= 64) ? KC_GEMM_K64 : (K >= 32) ? KC_GEMM_K32 : KC_GEMM_K16;
    // MAUS_LU_TRACE=<file>: one line "M N K batch" per trailing-update launch, in launch order (lets a PMC pass
    // attribute rocprofv3's per-dispatch counters to the K classes: tools/pmc_traffic.py)
    static FILE* trace = [] { const char* e = getenv("MAUS_LU_TRACE"); return e ? fopen(e, "a") : (FILE*)nullptr; }();
    if (trace) { fprintf(trace, "%d %d %d %d\n", M, N, K, w.G); fflush(trace); }
    prof(w, kc, 0);
    maus_zgemm_launch_lu(w.st, M, N, K, w.H, w.U, w.H, w.npad, w.strideH, k0, k0, c0, w.G, w.perm + r0, w.npad);
    prof(w, kc, 1, 8.0 * M * N * K * w.G, 16.0 * ((double)M * K + (double)K * N + 2.0 * M * N) * w.G);
}

static void lu_trsm(const LuWs& w, int j, int k, int c_lo, int c_hi) {
    if (c_hi <= c_lo) return;
    // up to 128 rows: one launch of the register-resident MFMA solve (trsm_mfma_kernel); beyond that two halves with the
    // zgemm update between them
    if (k <= 128) {
        prof(w, KC_TRSM, 0);
        dim3 grid(((c_hi - c_lo) / 16 + 3) / 4, w.G);
#define TRSM_MFMA(NB) case NB: hipLaunchKernelGGL((trsm_mfma_kernel<NB>), grid, dim3(256), 0, w.st, w.H, w.U, w.strideH, w.perm, w.npad, j, c_lo, c_hi); break
        switch (k / 16) { TRSM_MFMA(1); TRSM_MFMA(2); TRSM_MFMA(3); TRSM_MFMA(4); TRSM_MFMA(5); TRSM_MFMA(6); TRSM_MFMA(7); TRSM_MFMA(8); }
#undef TRSM_MFMA
        prof(w, KC_TRSM, 1, 4.0 * k * k * (c_hi - c_lo) * w.G, 32.0 * k * (c_hi - c_lo) * w.G);
        return;
    }
    const int h = (k / 32) * 16;
    lu_trsm(w, j, h, c_lo, c_hi);
    lu_gemm(w, j + h, j + k, c_lo, c_hi, j, j + h);
    lu_trsm(w, j + h, k - h, c_lo, c_hi);
}

static void lu_panel(const LuWs& w, int j0) {
    int m = w.npad - j0;
    prof(w, KC_PANEL, 0);
    dim3 grid(w.G), block(PT);
    int rpt = (m + PT - 1) / PT;
#define PANEL_IP(R, W) hipLaunchKernelGGL((lu_panel_ip_kernel<R, W>), grid, block, 0, w.st, w.H, w.U, (long)LU_TW, w.strideH, j0, m, w.ipiv, w.perm, w.npad, w.info)
    // Small batches: several workgroups per matrix (lu_panel_mw_kernel).  Only when the caller guarantees that this is the
    // only LU in flight on the device (w.mw_sync set) and all G*W workgroups fit on the chip at once -- the workgroups of a
    // matrix wait for each other.
#if MAUS_NBP == 16
    // m <= 1024: the whole slice in the registers of one workgroup
    if (m <= 2 * PT) {
#define PANEL_RS(NT) hipLaunchKernelGGL((lu_panel_rs_kernel<2, NT>), grid, dim3(NT), 0, w.st, w.H, w.U, (long)LU_TW, w.strideH, j0, m, w.ipiv, w.perm, w.npad, w.info)
        if (m <= 256) PANEL_RS(128); else if (m <= 512) PANEL_RS(256); else PANEL_RS(512);
#undef PANEL_RS
        prof(w, KC_PANEL, 1, 8.0 * m * NBP * NBP / 2 * w.G, 16.0 * m * NBP * 2 * w.G);
        return;
    }
    if (w.mw_sync && m > 1024) {
        static const int ncu = [] { int v = 0; int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev); return v > 0 ? v : 256; }();
        auto p2floor = [](int x) { int p = 1; while (2 * p <= x) p *= 2; return p; };
        auto p2ceil = [](int x) { int p = 1; while (p < x) p *= 2; return p; };
        const int wmin = p2ceil((m + 2 * PT - 1) / (2 * PT));
        const int wmax = std::min(p2floor(std::max(1, ncu / std::max(1, w.G))), MW_MAXW);
        const int W = std::max(wmin, std::min(wmax, p2floor(m / 256)));
        if (W >= 2 && W <= wmax) {
            const int rpt1 = (m + W * PT - 1) / (W * PT);
            const unsigned tag_base = (unsigned)(j0 / NBP + 1) << 5;
            if (rpt1 <= 1) hipLaunchKernelGGL((lu_panel_mw_kernel<1>), dim3(w.G, W), block, 0, w.st, w.H, w.U, (long)LU_TW, w.strideH, j0, m, W, w.ipiv, w.perm, w.npad, w.info, (MwSync*)w.mw_sync, w.mw_timeout, w.mw_force_abort, tag_base);
            else hipLaunchKernelGGL((lu_panel_mw_kernel<2>), dim3(w.G, W), block, 0, w.st, w.H, w.U, (long)LU_TW, w.strideH, j0, m, W, w.ipiv, w.perm, w.npad, w.info, (MwSync*)w.mw_sync, w.mw_timeout, w.mw_force_abort, tag_base);
            prof(w, KC_PANEL, 1, 8.0 * m * NBP * NBP / 2 * w.G, 16.0 * m * NBP * 2 * w.G);
            return;
        }
    }
#endif
    // 8-column register sub-blocks where the rows per thread allow it (m <= 2048): 24 instead of 40 column reads per panel
    // (33.0 vs 34.4 ms per 181-solve sweep, 37.5 vs 39.1 at 256, against 4-column sub-blocks everywhere)
    if (rpt <= 1) PANEL_IP(1, 8); else if (rpt <= 2) PANEL_IP(2, 8); else if (rpt <= 4) PANEL_IP(4, 8);
    else if (rpt <= 8) PANEL_IP(8, 4); else PANEL_IP(16, 2);
#undef PANEL_IP
    prof(w, KC_PANEL, 1, 8.0 * m * NBP * NBP / 2 * w.G, 16.0 * m * NBP * 8 * w.G);
}

static void lu_recurse(const LuWs& w, int j0, int wd) {
    if (wd <= NBP) { lu_panel(w, j0); return; }
    int h = (wd / (2 * NBP)) * NBP;
    lu_recurse(w, j0, h);
    lu_trsm(w, j0, h, j0 + h, j0 + wd);
    lu_gemm(w, j0 + h, w.npad, j0 + h, j0 + wd, j0, j0 + h);
    lu_recurse(w, j0 + h, wd - h);
}

#ifdef MAUS_PANEL_CLOCK
extern "C" int maus_debug_panel_clocks(unsigned long long* out, int reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_panel_clk), sizeof(z)) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_panel_clk), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif
// Maximum rows the base panel can own (512 threads x 16 rows per thread in the 2-column sub-block variant)
int maus_lu_max_npad() { return PT * 16; }
#if MAUS_NBP == 16
size_t maus_lu_mw_sync_bytes() { return sizeof(MwSync); }
#else
size_t maus_lu_mw_sync_bytes() { return 64; }
#endif

// Factor all G matrices in the workspace and carry the augmented column through (L y = P b).
void maus_lu_factor(const LuWs& w, int nbo) {
    const int ncols = (int)w.ldh;                 // npad + 32
    if (w.mw_sync) (void)hipMemsetAsync(w.mw_sync, 0, maus_lu_mw_sync_bytes() * (size_t)w.G, w.st);      // abort words and tags of the multi-workgroup panel
    hipLaunchKernelGGL(init_perm_kernel, dim3((w.npad + 255) / 256, w.G), dim3(256), 0, w.st, w.perm, w.npad);
    for (int J = 0; J < w.npad; J += nbo) {
        int wd = (w.npad - J < nbo) ? (w.npad - J) : nbo;
        lu_recurse(w, J, wd);
        lu_trsm(w, J, wd, J + wd, ncols);
        lu_gemm(w, J + wd, w.npad, J + wd, ncols, J, J + wd);
    }
}

void maus_lu_backsolve(const LuWs& w, c128* Wpop, long ldw, const int* d_slots, c128* xout_dense) {
    prof(w, KC_BACKSOLVE, 0);
    size_t shm = sizeof(c128) * ((size_t)w.npad + BSB * (BSB + 1) + 2 * BSB);
    static bool attr_set = false;
    if (!attr_set) {   // ~83 KB of dynamic LDS at npad = 4096, 145 KB at 8192 (160 KB per CU on gfx950)
        (void)hipFuncSetAttribute((const void*)backsolve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        attr_set = true;
    }
    // Few matrices (round 4, later): one workgroup per matrix draws a single CU's share of the bandwidth -- 134 MB of U per
    // matrix at ~40 GB/s, 3.3 ms whatever the batch.  Then the substitution runs in blocks of 256 columns: the block's own
    // triangle by the kernel above (x back into the augmented column), everything above it as ONE product over the whole
    // batch, y[0 : b0] -= U[0 : b0, block] x_block on the skinny zgemm (N = 1: the chip's bandwidth instead of G CUs').
    constexpr int BB = 256;                                  // (128 .. 1024 measured: 2.2-2.4 ms at 32 solves either way)
    if (w.G <= 64 && w.npad >= 2048 && (w.npad % BB) == 0 && w.ident != nullptr) {
        for (int b0 = w.npad - BB; b0 >= 0; b0 -= BB) {
            hipLaunchKernelGGL(backsolve_kernel, dim3(w.G), dim3(512), shm, w.st, w.U, (long)LU_TW, w.strideH, w.n, w.npad,
                               Wpop, ldw, d_slots, xout_dense, w.flags, b0, b0 + BB);
            if (b0 > 0) maus_zgemm_launch_lu(w.st, b0, 1, BB, w.U, w.U, w.U, w.npad, w.strideH, b0, b0, w.npad, w.G, w.ident, 0);
        }
        hipLaunchKernelGGL(backsolve_out_kernel, dim3((w.n + 255) / 256, w.G), dim3(256), 0, w.st, w.U, (long)LU_TW, w.strideH, w.n, w.npad,
                           Wpop, ldw, d_slots, xout_dense, w.flags);
    } else
    hipLaunchKernelGGL(backsolve_kernel, dim3(w.G), dim3(512), shm, w.st, w.U, (long)LU_TW, w.strideH, w.n, w.npad,
                       Wpop, ldw, d_slots, xout_dense, w.flags, 0, w.npad);
    prof(w, KC_BACKSOLVE, 1, 4.0 * w.npad * w.npad * w.G, 8.0 * w.npad * w.npad * w.G);
}

void maus_build_h(const LuWs& w, const c128* A, const c128* d_shift, const double* d_psi, int rhs_mode,
                  const c128* X, long ldx, const int* d_slots, const c128* bvec, int pert_mode, const double* d_U, int tiled) {
    prof(w, KC_BUILD, 0);
    hipLaunchKernelGGL(build_h_kernel, dim3(w.npad, w.G), dim3(256), 0, w.st, A, w.n, w.npad, w.ldh, w.strideH, w.H,
                       d_shift, d_psi, rhs_mode, X, ldx, d_slots, bvec, pert_mode, d_U, w.flags, tiled);
    prof(w, KC_BUILD, 1, 0, 32.0 * w.npad * w.ldh * w.G);
}

void maus_load_h(const LuWs& w, const c128* d_Ain, const c128* d_bin) {
    hipLaunchKernelGGL(load_h_kernel, dim3(w.npad, w.G), dim3(256), 0, w.st, d_Ain, d_bin, w.n, w.npad, w.ldh, w.strideH, w.H, w.flags);
}
