// Hermitian eigendecomposition on the device (gfx950).
//
// Replaces the reference's once-per-candidate
//     eigvals_h, eigvecs_h = scipy.linalg.eigh(A)                                  AMS:161
// (LAPACK zheevr: zhetrd -> dstemr -> zunmtr; one decomposition per matrix here), which at n = 8192 costs 73 s
// on the GPU box's host against 7 ms per loop body on the device:
//   1. maus_herm_tridiag        A = Q T Q^H, zhetrd('L') semantics: blocked Householder tridiagonalisation, zlatrd panels of
//                               64 columns + a rank-128 update of the trailing matrix per panel ([V W][W V]^H on the zgemm).
//                               Half of the 16/3 n^3 flops are Hermitian matrix-vector products with the trailing matrix --
//                               HBM-bound: 8/3 n^3 bytes since round 4, when the matvec, the update and everything else began
//                               to touch the lower triangle only (16/3 n^3 with the full storage of rounds 2-3) -- the other
//                               half the MFMA zgemm.
//   2. maus_herm_tridiag_eig    the real symmetric tridiagonal eigenproblem, O(n^2) per sweep: bisection on the Sturm count and
//                               eigenvectors from the twisted factorisation, one thread per eigenpair; the caller
//                               (engine.device_eigh) falls back to scipy.linalg.eigh_tridiagonal (LAPACK dstemr, the kernel
//                               zheevr uses) for clustered spectra
//   3. maus_herm_backtransform  V = Q Z, zunmtr semantics: blocks of 64 reflectors as I - V T V^H (zlarft), three zgemm
//                               calls per block.
// Phase convention: the reflectors are LAPACK's (zlarfg: beta real, v(1) = 1; H(i) acts on rows i+1..n-1), so Q e_1 = e_1
// and the first row of V is the first row of the real Z -- LAPACK's "first component real".  The SIGN of a column is
// whatever dstemr gives for T, and that is not reproducible across tridiagonalisations: dstemr fixes the sign at the twist
// index of its factorisation, which moves under a 1-ulp change of T (one column in ten flips between LAPACK's own T and
// the same T perturbed in the last bit, tests/test_gpu_herm_eigh.py).  Nothing downstream sees it: the distinctness and
// redundancy tests take |<v, s>| (AMS:436, 515), a candidate converged through the shortcut is not stepped again.
//
// Per column i of a panel (m' = n-i-1) three launches (round 4; six in round 3 -- 49 000 launches of 4-9 us at n = 8192 were
// the second largest cost of the reduction and more dispatches in flight than a counter-collecting profiler could take):
//   col1   a(i:n,i) -= V(i:n,0:j) conj(W(i,0:j)) + W(i:n,0:j) conj(V(i,0:j))            (row-parallel; partial |.|^2; d(i))
//   col2   every workgroup repeats zlarfg's scalar part from the partials in fixed order (beta / tau / scale; workgroup 0 keeps
//          e(i), tau(i)) and forms v = scale * a on the fly; then, by workgroup index,
//            hemv:  partial vectors of w0 = A22 v from up to eight 64 x 64 tiles of the LOWER triangle (each tile serves its rows
//                   and its columns), v -> panel column j and the reflector store, partial w0^H v
//                   (A22 = the trailing matrix as the previous panels left it: this panel's reflectors come in below)
//            dots:  t1 = W(:,0:j)^H v, t2 = V(:,0:j)^H v, one column each
//   col3   w0 from the partial vectors; u = w0 - V t1 - W t2,  w = tau u + alpha v  -> panel column j of W,  alpha = -1/2 tau (w^H v) with
//          w^H v = conj(tau) (w0^H v - 2 Re sum_k conj(t1_k) t2_k)  from the partials and the dots: no second pass over w;
//          rows <= i of a panel column are neither written nor read
// All reductions run in a fixed order (no atomics), so a decomposition is reproducible bit for bit.  The host keeps at most two
// panels of launches in flight.
#include "ctx.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>

void maus_zgemm_launch(hipStream_t st, int M, int N, int K, const c128* A, long lda, long sA,
                       const c128* B, long ldb, long sB, c128* C, long ldc, long sC,
                       double alpha, int beta, int batch, int blay, bool conja, bool conjb);

namespace {

constexpr int HNB = 64;        // panel width / reflectors per block
constexpr int HT = 256;        // threads of the row-parallel kernels

__device__ __forceinline__ double block_sum_d(double v, double* sbuf) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) sbuf[wave] = v;
    __syncthreads();
    double s = 0.0;
    for (int q = 0; q < nw; ++q) s += sbuf[q];
    return s;
}

// a(i:n, i) with the panel's earlier reflectors applied; col[r] for r in [i, n); part[block] = sum |a(r)|^2 over its rows >= i+2.
// 64 rows per workgroup, the 2 j terms of a row split over its four waves (partial sums combined in a fixed order): with a
// whole row's terms in one thread the launch was 32 workgroups of 126 dependent loads each, 18 us per column at n = 8192.
constexpr int HR = 64;         // rows per workgroup of col1 / col3
__global__ void __launch_bounds__(256)
herm_col1_kernel(const c128* __restrict__ Aw, int n, int i, int j, const c128* __restrict__ PV, const c128* __restrict__ PW,
                 c128* __restrict__ col, double* __restrict__ part, double* __restrict__ d)
{
    __shared__ c128 sw[HNB], sv[HNB];
    __shared__ c128 sp[4][HR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < j) { sw[tid] = cconj(PW[(long)tid * n + i]); sv[tid] = cconj(PV[(long)tid * n + i]); }
    __syncthreads();
    const int r = i + blockIdx.x * HR + lane;
    const int kc = (j + 3) / 4, k0 = wave * kc, k1 = min(j, k0 + kc);
    c128 p = cmake(0.0, 0.0);
    if (r < n) {
        // four columns of V and of W requested together: with one pair per iteration the loop waited out a memory latency per term
        int k = k0;
        for (; k + 4 <= k1; k += 4) {
            c128 a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = PV[(long)(k + u) * n + r]; b[u] = PW[(long)(k + u) * n + r]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { cfma(p, a[u], sw[k + u]); cfma(p, b[u], sv[k + u]); }
        }
        for (; k < k1; ++k) { cfma(p, PV[(long)k * n + r], sw[k]); cfma(p, PW[(long)k * n + r], sv[k]); }
    }
    sp[wave][lane] = p;
    __syncthreads();
    if (wave == 0) {
        double ss = 0.0;
        if (r < n) {
            const c128 q = cadd(cadd(sp[0][lane], sp[1][lane]), cadd(sp[2][lane], sp[3][lane]));
            const c128 a = csub(Aw[(long)r * n + i], q);
            col[r] = a;
            if (r == i) d[i] = a.x;
            if (r >= i + 2) ss = fma(a.x, a.x, a.y * a.y);
        }
        ss = wave_sum(ss);
        if (lane == 0) part[blockIdx.x] = ss;
    }
}

// zlarfg's scalars for (alpha = col[i+1], x = col[i+2:n]); ss = sum |x|^2
struct Larfg { c128 tau, scal; double beta; };
__device__ __forceinline__ Larfg herm_larfg_scalars(const c128* __restrict__ col, int i, double ss)
{
    const double xnorm = sqrt(ss);
    const c128 alpha = col[i + 1];
    Larfg L;
    L.tau = cmake(0.0, 0.0); L.scal = cmake(0.0, 0.0); L.beta = alpha.x;
    if (!(xnorm == 0.0 && alpha.y == 0.0)) {
        const double w = fmax(fmax(fabs(alpha.x), fabs(alpha.y)), xnorm);        // dlapy3
        const double ax = alpha.x / w, ay = alpha.y / w, xn = xnorm / w;
        const double nrm = w * sqrt(ax * ax + ay * ay + xn * xn);
        L.beta = -copysign(nrm, alpha.x);
        L.tau = cmake((L.beta - alpha.x) / L.beta, -alpha.y / L.beta);
        L.scal = crecip(cmake(alpha.x - L.beta, alpha.y));
    }
    return L;
}

// The Hermitian matvec w0 = A22 v reads the LOWER triangle only (round 4; rounds 2-3 read full rows of a matrix kept in both
// triangles: 16/3 n^3 bytes over the reduction, 543 of its 710 ms at n = 8192).  The trailing matrix is cut into 64 x 64 tiles
// on the absolute grid; a tile (bi, bj), bi >= bj, serves its rows (u_R += A[R][c] v_c) and its columns (z_c += conj(A[R][c]) v_R),
// the diagonal tile with its lower part only and a real diagonal (zhemv 'L').  A workgroup of eight waves takes up to hch tiles (4 .. 16, chosen per column)
// of one block row; wave w owns the columns 8 w .. 8 w + 7 of every tile and all 64 rows (a load instruction = eight rows x one
// 128-byte line), so the column sums z are complete inside the wave -- a register sum over the eight loads, then over the
// eight row-lanes -- and leave per tile without a barrier; the row sums u stay in registers over the whole chunk and cross
// the waves once at its end.  Nothing is accumulated across workgroups: the partial vectors go to PU[chunk][row] and
// PZ[block row][column], and col3 adds them in a fixed order,  w0_r = sum_ch PU[ch][r] + sum_{bi >= block(r)} PZ[bi][r]
// (6 % of the tile traffic).
// (Measured on the way: one wave per tile with lane = column and a DPP reduction per row, 653 ms of col2 at n = 8192; eight-row
// slices per wave with an LDS exchange and two barriers per tile, 488 ms.)
// Workgroups [0, ns): the tiles; also v -> PV[j], VQ[i] (first chunk of a block row) and cpart[b] = sum conj(partial) v over
// the workgroup's partials (w0^H v is linear in them).
// Workgroups [ns, ns + 2j): t[k] = W_k^H v (k < j), t[j + k] = V_k^H v over the rows > i.
constexpr int HTB = 64;        // tile edge
constexpr int HCH_MIN = 4, HCH_MAX = 16;   // tiles per workgroup: chosen per column so that the launch has ~3 workgroups per CU
constexpr int HW2 = 8;         // waves of a col2 workgroup

#define MAUS_DPP_F64(V, CTRL) __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(V), CTRL, 0xf, 0xf, false), \
                                               __builtin_amdgcn_update_dpp(0, __double2loint(V), CTRL, 0xf, 0xf, false))
// sum over the eight lanes that differ in the low three bits of the lane number; every one of them gets it
__device__ __forceinline__ double sum_low8(double v) {
    v += MAUS_DPP_F64(v, 0xB1);            // quad_perm [1,0,3,2]
    v += MAUS_DPP_F64(v, 0x4E);            // quad_perm [2,3,0,1]
    v += MAUS_DPP_F64(v, 0x141);           // row_half_mirror: the other quad of the eight
    return v;
}
// sum over the eight lanes l, l + 8, .., l + 56 (l < 8), valid in lanes 0..7: the two halves of a 16-lane row by a rotation,
// the four rows through the permute network in a fixed order
__device__ __forceinline__ double sum_stride8(double v) {
    v += MAUS_DPP_F64(v, 0x128);           // row_ror:8
    v += __shfl_down(v, 32, 64);
    v += __shfl_down(v, 16, 64);
    return v;
}

__global__ void __launch_bounds__(64 * HW2, 4)
herm_col2_kernel(const c128* __restrict__ Aw, int n, int i, int j, int ns, int hch, const c128* __restrict__ col, const double* __restrict__ part,
                 int nparts, c128* __restrict__ PV, const c128* __restrict__ PW, c128* __restrict__ VQ, c128* __restrict__ tau,
                 double* __restrict__ e, c128* __restrict__ PU, c128* __restrict__ PZ, c128* __restrict__ cpart, c128* __restrict__ t)
{
    __shared__ c128 s_scal;
    __shared__ double sbuf[HW2];
    __shared__ c128 s_dot[HW2];
    __shared__ c128 s_v[HTB];
    __shared__ c128 s_u[HW2][HTB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const c128 zero = cmake(0.0, 0.0);
    // (block row, chunk) of a tile workgroup: block row b0 + tb has tb / hch + 1 chunks of hch tiles
    const int b0 = (i + 1) / HTB;
    int bi = 0, ch = 0, bj0 = 0, bj1 = -1;
    if (b < ns) {
        int tb = 0, base = 0;
        for (;;) { const int nc = tb / hch + 1; if (b < base + nc) break; base += nc; ++tb; }
        ch = b - base; bi = b0 + tb;
        bj0 = b0 + ch * hch; bj1 = min(bi, bj0 + hch - 1);
    }
    const int r0 = bi * HTB;
    const int rr = lane >> 3, cc = lane & 7;                              // row inside a group of eight, column inside the wave's eight
    // 32 rows of a slice at a time: four loads in a lane's registers, the next four in flight
    c128 a[4];
    auto load = [&](int bj, int half, c128 (&x)[4]) {                     // rows 32 half .. + 31 of this wave's 64 x 8 slice of tile (bi, bj)
        const int c = bj * HTB + 8 * wave + cc;
        const bool cok = c > i && c < n;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int R = r0 + 32 * half + 8 * k + rr;
            x[k] = (cok && R > i && R < n) ? Aw[(long)R * n + c] : zero;
        }
    };
    // the first rows are requested before the scalars below exist: they do not depend on them
    if (bj0 <= bj1) load(bj0, 0, a);
    if (wave == 0) {
        // the |.|^2 partials of col1 in a fixed order: lane-strided, then the butterfly -- the same bits in every workgroup
        double ss = 0.0;
        for (int q = lane; q < nparts; q += 64) ss += part[q];
        ss = wave_sum(ss);
        if (lane == 0) {
            const Larfg L = herm_larfg_scalars(col, i, ss);
            s_scal = L.scal;
            if (blockIdx.x == 0) { tau[i] = L.tau; e[i] = L.beta; }
        }
    }
    lds_barrier();
    const c128 scal = s_scal;
    auto vat = [&](int c) { return (c == i + 1) ? cmake(1.0, 0.0) : cmul(col[c], scal); };     // c > i
    if (b < ns) {
        if (tid < HTB) {
            const int R = r0 + tid;
            const bool on = R > i && R < n;
            const c128 v = on ? vat(R) : zero;
            s_v[tid] = v;
            if (ch == 0 && on) { PV[(long)j * n + R] = v; VQ[(long)i * n + R] = v; }
        }
        lds_barrier();
        c128 acc0[4], acc1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc0[k] = acc1[k] = zero;
        c128 zdot = zero;
        for (int bj = bj0; bj <= bj1; ++bj) {
            const int c = bj * HTB + 8 * wave + cc;
            const bool cok = c > i && c < n;
            const c128 vc = cok ? vat(c) : zero;
            const bool diag = bj == bi;
            c128 zc = zero;
            auto half = [&](int h, c128 (&acc)[4]) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int rl = 32 * h + 8 * k + rr, R = r0 + rl;
                    c128 x = a[k];
                    if (diag) {
                        if (R == c) x.y = 0.0;
                        if (R >= c) cfma(acc[k], x, vc);
                        if (R > c) cfma_conj(zc, x, s_v[rl]);
                    } else {
                        cfma(acc[k], x, vc);
                        cfma_conj(zc, x, s_v[rl]);
                    }
                }
            };
            c128 an[4];
            load(bj, 1, an);
            half(0, acc0);
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = an[k];
            if (bj < bj1) load(bj + 1, 0, an);
            half(1, acc1);
            if (bj < bj1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) a[k] = an[k];
            }
            const c128 z = cmake(sum_stride8(zc.x), sum_stride8(zc.y));   // lanes 0..7: the column sums of this wave's columns
            if (lane < 8) {
                if (c < n) PZ[(long)bi * n + c] = z;
                cfma_conj(zdot, z, vc);                                    // vc = 0 where the column is not part of the matvec
            }
        }
        // the row sums: over the wave's eight columns on the DPP network, over the waves through LDS
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double sr0 = sum_low8(acc0[k].x), si0 = sum_low8(acc0[k].y), sr1 = sum_low8(acc1[k].x), si1 = sum_low8(acc1[k].y);
            if (cc == 0) { s_u[wave][8 * k + rr] = cmake(sr0, si0); s_u[wave][32 + 8 * k + rr] = cmake(sr1, si1); }
        }
        const double zr = wave_sum(lane < 8 ? zdot.x : 0.0), zi = wave_sum(lane < 8 ? zdot.y : 0.0);
        if (lane == 0) s_dot[wave] = cmake(zr, zi);
        lds_barrier();
        if (wave == 0) {
            c128 u = s_u[0][lane];
#pragma unroll
            for (int q = 1; q < HW2; ++q) u = cadd(u, s_u[q][lane]);
            const int R = r0 + lane;
            if (R < n) PU[(long)ch * n + R] = u;
            c128 d = zero;
            cfma_conj(d, u, s_v[lane]);                                    // s_v = 0 on the rows that are not part of the matvec
            const double dr = wave_sum(d.x), di = wave_sum(d.y);
            if (lane == 0) {
                c128 dsum = cmake(dr, di);
                for (int q = 0; q < HW2; ++q) dsum = cadd(dsum, s_dot[q]);
                cpart[b] = dsum;
            }
        }
    } else {
        const int k = b - ns;
        const c128* p = (k < j) ? PW + (long)k * n : PV + (long)(k - j) * n;
        c128 s = cmake(0.0, 0.0);
        for (int r = i + 1 + tid; r < n; r += 64 * HW2) cfma_conj(s, p[r], vat(r));
        const double sr = block_sum_d(s.x, sbuf), si = block_sum_d(s.y, sbuf);
        if (tid == 0) t[k] = cmake(sr, si);
    }
}
#undef MAUS_DPP_F64

// u = w0 - V (W^H v) - W (V^H v), w = tau u + alpha v -> panel column j of W, with w0 summed from the partial vectors of
// col2 (fixed order: the chunks of the row's block, then the block rows from its own downwards).  One 64-row block of the
// tile grid per workgroup; the terms of a row are split over the four waves as in col1.
__global__ void __launch_bounds__(256)
herm_col3_kernel(const c128* __restrict__ PV, c128* __restrict__ PW, int n, int i, int j, const c128* __restrict__ PU,
                 const c128* __restrict__ PZ, const c128* __restrict__ t, const c128* __restrict__ tau, const c128* __restrict__ cpart, int ns, int hch)
{
    __shared__ c128 st[2 * HNB];
    __shared__ double sbuf[4];
    __shared__ c128 s_alpha;
    __shared__ c128 sp[4][HR];
    __shared__ c128 sw[4][HR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 2 * j) st[tid] = t[tid];
    // u^H v = w0^H v - 2 Re sum_k conj(t1_k) t2_k: the partials of col2 and, from thread k < j, its term of the correction
    c128 ds = cmake(0.0, 0.0);
    for (int q = tid; q < ns; q += 256) ds = cadd(ds, cpart[q]);
    if (tid < j) { const c128 a = t[tid], b = t[j + tid]; ds.x -= 2.0 * (a.x * b.x + a.y * b.y); }
    const double dr = block_sum_d(ds.x, sbuf), di = block_sum_d(ds.y, sbuf);       // (the barriers inside also publish st)
    if (tid == 0) {
        const c128 tq = tau[i];
        const c128 whv = cmul(cconj(tq), cmake(dr, di));                         // w^H v
        const c128 th = cmul(tq, whv);
        s_alpha = cmake(-0.5 * th.x, -0.5 * th.y);
    }
    const int b0 = (i + 1) / HTB, nbk = (n + HTB - 1) / HTB;
    const int B = b0 + blockIdx.x;
    const int r = B * HTB + lane;
    const bool on = r > i && r < n;                                              // rows <= i of a panel column are never read
    const int qc = (2 * j + 3) / 4, q0 = wave * qc, q1 = min(2 * j, q0 + qc);
    c128 p = cmake(0.0, 0.0), ws = cmake(0.0, 0.0);
    if (on) {
        auto term = [&](int q) { return (q < j) ? PV[(long)q * n + r] : PW[(long)(q - j) * n + r]; };
        int q = q0;
        for (; q + 8 <= q1; q += 8) {                       // eight terms in flight (see col1)
            c128 a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = term(q + u);
#pragma unroll
            for (int u = 0; u < 8; ++u) cfma(p, a[u], st[q + u]);
        }
        for (; q < q1; ++q) cfma(p, term(q), st[q]);
        // the partial vectors of the matvec
        const int ncu = (B - b0) / hch + 1, nt = ncu + (nbk - B);
        const int tc = (nt + 3) / 4, t0 = wave * tc, t1 = min(nt, t0 + tc);
        auto pv = [&](int q) { return (q < ncu) ? PU[(long)q * n + r] : PZ[(long)(B + q - ncu) * n + r]; };
        q = t0;
        for (; q + 8 <= t1; q += 8) {
            c128 a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = pv(q + u);
#pragma unroll
            for (int u = 0; u < 8; ++u) ws = cadd(ws, a[u]);
        }
        for (; q < t1; ++q) ws = cadd(ws, pv(q));
    }
    sp[wave][lane] = p;
    sw[wave][lane] = ws;
    __syncthreads();
    if (wave != 0 || !on) return;
    const c128 w0 = cadd(cadd(sw[0][lane], sw[1][lane]), cadd(sw[2][lane], sw[3][lane]));
    const c128 u = csub(w0, cadd(cadd(sp[0][lane], sp[1][lane]), cadd(sp[2][lane], sp[3][lane])));
    c128 w = cmul(tau[i], u);
    cfma(w, s_alpha, PV[(long)j * n + r]);
    PW[(long)j * n + r] = w;
}

// rows r0..n-1 of the panel as GEMM operands: P2[r][0:nb] = V, P2[r][nb:2nb] = W; Q2[r] = [W | V]   (row-major, ld = 2 nb)
__global__ void __launch_bounds__(HT)
herm_pack_kernel(const c128* __restrict__ PV, const c128* __restrict__ PW, int n, int r0, int nb, c128* __restrict__ P2, c128* __restrict__ Q2)
{
    const long e = (long)blockIdx.x * HT + threadIdx.x;            // over (n - r0) x nb, row index fastest: coalesced panel reads
    const int m = n - r0;
    if (e >= (long)m * nb) return;
    const int k = (int)(e / m), rr = (int)(e - (long)k * m);
    const c128 a = PV[(long)k * n + r0 + rr], b = PW[(long)k * n + r0 + rr];
    c128* p = P2 + (long)rr * 2 * nb;
    c128* q = Q2 + (long)rr * 2 * nb;
    p[k] = a; p[nb + k] = b; q[k] = b; q[nb + k] = a;
}

// Vt[rr][k] = VQ[k0 + k][r0 + rr]: the block's reflectors as a row-major (n - r0) x nb operand
__global__ void __launch_bounds__(HT)
herm_vt_kernel(const c128* __restrict__ VQ, int n, int k0, int r0, int nb, c128* __restrict__ Vt)
{
    const long e = (long)blockIdx.x * HT + threadIdx.x;
    const int m = n - r0;
    if (e >= (long)m * nb) return;
    const int k = (int)(e / m), rr = (int)(e - (long)k * m);
    Vt[(long)rr * nb + k] = VQ[(long)(k0 + k) * n + r0 + rr];
}

// zlarft (forward, columnwise): T upper triangular nb x nb (row-major, ld = HNB) from G = V^H V and tau
__global__ void __launch_bounds__(64)
herm_larft_kernel(const c128* __restrict__ G, int nb, const c128* __restrict__ tau, c128* __restrict__ T)
{
    __shared__ c128 sT[HNB][HNB];                                  // 64 KB
    const int ii = threadIdx.x;                                    // row of T
    for (int c = 0; c < nb; ++c) sT[ii][c] = cmake(0.0, 0.0);
    __syncthreads();
    for (int jj = 0; jj < nb; ++jj) {
        const c128 tj = tau[jj];
        if (ii < jj) {
            c128 s = cmake(0.0, 0.0);
            for (int l = ii; l < jj; ++l) cfma(s, sT[ii][l], G[(long)l * HNB + jj]);
            const c128 m = cmul(tj, s);
            sT[ii][jj] = cmake(-m.x, -m.y);
        } else if (ii == jj) sT[ii][jj] = tj;
        __syncthreads();
    }
    if (ii < nb) for (int c = 0; c < nb; ++c) T[(long)ii * HNB + c] = sT[ii][c];
}

// The work copy as zheevr('L') sees the matrix: only the lower triangle counts.  The upper triangle becomes its mirror image and
// the diagonal real, so that the full-row matvecs above compute what LAPACK's zhemv('L') computes -- also for an input that
// is Hermitian only to np.allclose's tolerance (AMS:384: the reference's own test before it calls eigh).
__global__ void __launch_bounds__(256)
herm_mirror_lower_kernel(c128* __restrict__ Aw, int n)
{
    __shared__ c128 tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int bi = blockIdx.y, bj = blockIdx.x;                      // tile (bi, bj) of the UPPER triangle, bj >= bi
    if (bj < bi) return;
    for (int q = ty; q < 32; q += 8) {                               // read the mirror tile (bj, bi) of the lower triangle
        const int r = bj * 32 + q, c = bi * 32 + tx;
        tile[q][tx] = (r < n && c < n) ? Aw[(long)r * n + c] : cmake(0.0, 0.0);
    }
    __syncthreads();
    for (int q = ty; q < 32; q += 8) {
        const int r = bi * 32 + q, c = bj * 32 + tx;
        if (r < n && c < n) {
            if (c > r) Aw[(long)r * n + c] = cconj(tile[tx][q]);
            else if (c == r) Aw[(long)r * n + c].y = 0.0;
        }
    }
}

__global__ void __launch_bounds__(HT)
herm_real_to_complex_kernel(const double* __restrict__ Z, long count, c128* __restrict__ V)
{
    const long e = (long)blockIdx.x * HT + threadIdx.x;
    if (e < count) V[e] = cmake(Z[e], 0.0);
}

// V[r][k] = Zt[k][r] + 0i: LAPACK hands the eigenvectors of T back column-major, i.e. as the transpose in C order -- turned
// here through a 32 x 32 LDS tile instead of by a 512 MB transposition on the host (seconds at n = 8192)
__global__ void __launch_bounds__(256)
herm_transpose_to_complex_kernel(const double* __restrict__ Zt, int n, c128* __restrict__ V)
{
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int k0 = blockIdx.y * 32, r0 = blockIdx.x * 32;
    for (int q = ty; q < 32; q += 8) { const int k = k0 + q, r = r0 + tx; tile[q][tx] = (k < n && r < n) ? Zt[(long)k * n + r] : 0.0; }
    __syncthreads();
    for (int q = ty; q < 32; q += 8) { const int r = r0 + q, k = k0 + tx; if (r < n && k < n) V[(long)r * n + k] = cmake(tile[tx][q], 0.0); }
}

}  // namespace

extern "C" {

// MAUS_HERM_TIMING=1: host wall time of the phases on stderr (allocation of GiB-sized buffers is not free)
// ---- the real symmetric tridiagonal eigenproblem on the device (r03) ---------------------------------------------------------
// Eigenvalue i by bisection on the Sturm count (one thread per eigenvalue; LAPACK dstebz's recurrence and pivot guard), its
// eigenvector from the twisted factorisation of T - lambda I (Fernando / Parlett-Dhillon, the getvec step of dstemr): a forward
// L D+ L^T and a backward U D- U^T sweep, the twist at the index of the smallest |gamma_k| = |D+_k + D-_k - (d_k - lambda)|,
// z_r = 1 and two two-term recurrences away from it.  O(n) per vector, no reorthogonalisation: for an eigenvalue that the
// bisection knows to eps ||T|| the vector is off by eps ||T|| / gap, so the caller (engine.device_eigh) accepts the result only
// when the smallest gap is large against eps ||T|| and the residuals measured here are at rounding level, and otherwise takes
// the host's dstemr.  Arrays are [k][i] (component k of vector i): every access is coalesced over the eigenvalue index.
__global__ void __launch_bounds__(256)
tri_bisect_kernel(const double* __restrict__ d, const double* __restrict__ e2, int n, double gl, double gu, double pivmin,
                  double* __restrict__ w)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double lo = gl, hi = gu;
    for (int it = 0; it < 1100; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (!(mid > lo && mid < hi)) break;                       // the interval is one ulp wide
        double q = d[0] - mid;
        if (fabs(q) < pivmin) q = -pivmin;
        int cnt = q < 0.0;
        for (int k = 1; k < n; ++k) {
            q = (d[k] - mid) - e2[k - 1] / q;
            if (fabs(q) < pivmin) q = -pivmin;
            cnt += q < 0.0;
        }
        if (cnt > i) hi = mid; else lo = mid;                     // cnt = number of eigenvalues below mid
    }
    w[i] = 0.5 * (lo + hi);
}

__global__ void __launch_bounds__(256)
tri_twisted_kernel(const double* __restrict__ d, const double* __restrict__ e, int n, const double* __restrict__ w, double tiny,
                   double* __restrict__ Z, double* __restrict__ B, double* __restrict__ nrm2)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long ld = n;
    const double lam = w[i];
    auto guard = [tiny](double x) { return fabs(x) < tiny ? copysign(tiny, x) : x; };
    double dp = guard(d[0] - lam);                                // forward: D+_k into Z[k][i]
    Z[i] = dp;
    for (int k = 0; k + 1 < n; ++k) {
        const double l = e[k] / dp;
        dp = guard((d[k + 1] - lam) - l * e[k]);
        Z[(long)(k + 1) * ld + i] = dp;
    }
    double dm = guard(d[n - 1] - lam);                            // backward: D-_k into B[k][i]; the twist index on the way
    B[(long)(n - 1) * ld + i] = dm;
    double best = fabs(dp + dm - (d[n - 1] - lam));
    int r = n - 1;
    for (int k = n - 2; k >= 0; --k) {
        const double u = e[k] / dm;
        dm = guard((d[k] - lam) - u * e[k]);
        B[(long)k * ld + i] = dm;
        const double g = fabs(Z[(long)k * ld + i] + dm - (d[k] - lam));
        if (g < best) { best = g; r = k; }
    }
    double ss = 1.0, zk = 1.0;                                    // z_r = 1;  z_k = -(e_k / D+_k) z_{k+1} below the twist
    for (int k = r - 1; k >= 0; --k) {
        zk = -(e[k] / Z[(long)k * ld + i]) * zk;
        Z[(long)k * ld + i] = zk;
        ss = fma(zk, zk, ss);
    }
    zk = 1.0;                                                     // z_{k+1} = -(e_k / D-_{k+1}) z_k above it
    for (int k = r; k + 1 < n; ++k) {
        zk = -(e[k] / B[(long)(k + 1) * ld + i]) * zk;
        Z[(long)(k + 1) * ld + i] = zk;
        ss = fma(zk, zk, ss);
    }
    Z[(long)r * ld + i] = 1.0;
    nrm2[i] = ss;
}

// Z[k][i] /= ||z_i||, then res[i] = max_k |(T z_i - lambda_i z_i)_k| of the normalised vector
__global__ void __launch_bounds__(256)
tri_scale_kernel(double* __restrict__ Z, int n, const double* __restrict__ nrm2)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (i < n) Z[(long)k * n + i] *= 1.0 / sqrt(nrm2[i]);
}

__global__ void __launch_bounds__(256)
tri_resid_kernel(const double* __restrict__ d, const double* __restrict__ e, int n, const double* __restrict__ w,
                 const double* __restrict__ Z, double* __restrict__ res)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double lam = w[i];
    double zm = 0.0, z0 = Z[i], worst = 0.0;
    for (int k = 0; k < n; ++k) {
        const double zp = (k + 1 < n) ? Z[(long)(k + 1) * n + i] : 0.0;
        double rk = (d[k] - lam) * z0;
        if (k > 0) rk = fma(e[k - 1], zm, rk);
        if (k + 1 < n) rk = fma(e[k], zp, rk);
        worst = fmax(worst, fabs(rk));
        zm = z0; z0 = zp;
    }
    res[i] = worst;
}

struct HermClock {
    bool on; std::chrono::steady_clock::time_point t;
    HermClock() : on(getenv("MAUS_HERM_TIMING") && atoi(getenv("MAUS_HERM_TIMING"))), t(std::chrono::steady_clock::now()) {}
    void lap(const char* what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[maus_herm] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

static void herm_free(maus_ctx* c) {
    if (c->hz) { (void)hipFree(c->hz); c->hz = nullptr; c->hzn = 0; }
    if (c->hq) { (void)hipFree(c->hq); c->hq = nullptr; }
    if (c->htau) { (void)hipFree(c->htau); c->htau = nullptr; }
    c->hqn = 0;
}

// The reflector store (n x n) and what else the decomposition keeps between its stages: callers that only wanted the
// tridiagonal matrix (eigenvalues for the reporting prologue, singular values) hand it back at once -- 1 GiB at n = 8192 that
// would otherwise stay resident until the next maus_set_matrix.
int maus_herm_release(maus_ctx* c) {
    if (!c) return -1;
    HIPCHK(c, hipStreamSynchronize(c->st));
    herm_free(c);
    return 0;
}

int maus_herm_tridiag(maus_ctx* c, double* d_out, double* e_out) {
    if (!c->A || c->rows != c->cols) FAIL(c, "maus_herm_tridiag: square matrix required (maus_set_matrix)");
    if (!d_out || (!e_out && c->rows > 1)) FAIL(c, "maus_herm_tridiag: null output");
    const int n = c->rows;
    HermClock clk;
    herm_free(c);
    const size_t nn = (size_t)n * n;
    c128 *Aw = nullptr, *PV = nullptr, *PW = nullptr, *P2 = nullptr, *Q2 = nullptr, *col = nullptr, *PU = nullptr, *PZ = nullptr, *t = nullptr, *cpart = nullptr;
    double *part = nullptr, *d = nullptr, *e = nullptr;
    const int nparts_max = (n + HR - 1) / HR;
    const int nbk = (n + HTB - 1) / HTB, nch_max = (nbk + HCH_MIN - 1) / HCH_MIN;      // tile grid of the matvec
    auto strips = [](int T, int hch) { int s = 0; for (int tb = 0; tb < T; ++tb) s += tb / hch + 1; return s; };   // workgroups for T block rows
    auto cleanup = [&]() {
        void* ps[] = {Aw, PV, PW, P2, Q2, col, PU, PZ, t, cpart, part, d, e};
        for (void* p : ps) if (p) (void)hipFree(p);
    };
#define HERM_ALLOC(ptr, bytes) do { if (hipMalloc((void**)&(ptr), (bytes)) != hipSuccess) { (void)hipGetLastError(); cleanup(); herm_free(c); \
        FAIL(c, "maus_herm_tridiag: out of device memory"); } } while (0)
    HERM_ALLOC(Aw, sizeof(c128) * nn);
    HERM_ALLOC(c->hq, sizeof(c128) * nn);
    HERM_ALLOC(c->htau, sizeof(c128) * n);
    HERM_ALLOC(PV, sizeof(c128) * (size_t)HNB * n);
    HERM_ALLOC(PW, sizeof(c128) * (size_t)HNB * n);
    HERM_ALLOC(P2, sizeof(c128) * (size_t)2 * HNB * n);
    HERM_ALLOC(Q2, sizeof(c128) * (size_t)2 * HNB * n);
    HERM_ALLOC(col, sizeof(c128) * n);
    HERM_ALLOC(PU, sizeof(c128) * (size_t)nch_max * n);
    HERM_ALLOC(PZ, sizeof(c128) * (size_t)nbk * n);
    HERM_ALLOC(t, sizeof(c128) * 2 * HNB);
    HERM_ALLOC(cpart, sizeof(c128) * (size_t)strips(nbk, HCH_MIN));
    HERM_ALLOC(part, sizeof(double) * nparts_max);
    HERM_ALLOC(d, sizeof(double) * n);
    HERM_ALLOC(e, sizeof(double) * n);
#undef HERM_ALLOC
    c->hqn = n;
    clk.lap("tridiag: allocations");
    hipStream_t st = c->st;
    hipError_t err = hipMemcpyAsync(Aw, c->A, sizeof(c128) * nn, hipMemcpyDeviceToDevice, st);
    if (err == hipSuccess) hipLaunchKernelGGL(herm_mirror_lower_kernel, dim3((n + 31) / 32, (n + 31) / 32), dim3(256), 0, st, Aw, n);
    if (err == hipSuccess) err = hipMemsetAsync(c->htau, 0, sizeof(c128) * n, st);
    if (err == hipSuccess) err = hipMemsetAsync(c->hq, 0, sizeof(c128) * nn, st);
    if (err == hipSuccess) err = hipMemsetAsync(e, 0, sizeof(double) * n, st);
    hipEvent_t ev[2] = {nullptr, nullptr};
    for (auto& x : ev) if (err == hipSuccess) err = hipEventCreateWithFlags(&x, hipEventDisableTiming);
    int panel = 0;
    for (int i0 = 0; i0 < n && err == hipSuccess; i0 += HNB, ++panel) {
        const int nb = (n - i0 < HNB) ? n - i0 : HNB;
        // at most two panels (~400 launches) in flight
        if (panel >= 2) err = hipEventSynchronize(ev[panel & 1]);
        if (err != hipSuccess) break;
        for (int j = 0; j < nb; ++j) {
            const int i = i0 + j, m = n - i, mp = n - i - 1;
            const int np1 = (m + HR - 1) / HR;
            hipLaunchKernelGGL(herm_col1_kernel, dim3(np1), dim3(256), 0, st, Aw, n, i, j, PV, PW, col, part, d);
            if (mp <= 0) continue;
            const int T = nbk - (i + 1) / HTB;                              // block rows of the tile grid that hold rows > i
            int hch = HCH_MAX;                                              // T (T + 1) / 2 tiles over ~768 workgroups
            while (hch > HCH_MIN && strips(T, hch) < 768) hch /= 2;
            const int ns = strips(T, hch);
            hipLaunchKernelGGL(herm_col2_kernel, dim3(ns + 2 * j), dim3(64 * HW2), 0, st, Aw, n, i, j, ns, hch, col, part, np1, PV, PW, c->hq, c->htau, e,
                               PU, PZ, cpart, t);
            hipLaunchKernelGGL(herm_col3_kernel, dim3(T), dim3(256), 0, st, PV, PW, n, i, j, PU, PZ, t, c->htau, cpart, ns, hch);
        }
        const int r0 = i0 + nb, M = n - r0;
        if (M > 0) {
            // A22 -= V W^H + W V^H as the product [V W] [W V]^H (K = 2 nb) on the LOWER triangle, which is all the matvecs and
            // col1 read (round 4): column strips of 1024, each from its diagonal block downwards -- 56 % of the full square's
            // flops at M = 8192.  (Rounds 2-3 updated both triangles for matvecs that read full rows.)
            hipLaunchKernelGGL(herm_pack_kernel, dim3((unsigned)(((long)M * nb + HT - 1) / HT)), dim3(HT), 0, st, PV, PW, n, r0, nb, P2, Q2);
            constexpr int SW = 1024;
            for (int c0 = 0; c0 < M; c0 += SW) {
                const int w = (M - c0 < SW) ? M - c0 : SW, mr = M - c0;
                ProfScope ps(c, KC_GEMM, 8.0 * mr * (double)w * 2 * nb, 16.0 * ((double)(mr + w) * 2 * nb + 2.0 * mr * w));
                maus_zgemm_launch(st, mr, w, 2 * nb, P2 + (size_t)c0 * 2 * nb, 2 * nb, 0, Q2 + (size_t)c0 * 2 * nb, 2 * nb, 0,
                                  Aw + (size_t)(r0 + c0) * n + r0 + c0, n, 0, -1.0, 1, 1, 1, false, true);
            }
        }
        err = hipGetLastError();
        if (err == hipSuccess) err = hipEventRecord(ev[panel & 1], st);
    }
    for (auto& x : ev) if (x) (void)hipEventDestroy(x);
    if (err == hipSuccess && (maus_stage_d2h(c, d_out, d, sizeof(double) * n, st)
                              || (n > 1 && maus_stage_d2h(c, e_out, e, sizeof(double) * (n - 1), st)))) { cleanup(); herm_free(c); return -1; }
    if (err == hipSuccess) err = hipStreamSynchronize(st);
    if (err == hipSuccess) err = hipGetLastError();
    clk.lap("tridiag: enqueue + kernels");
    cleanup();
    clk.lap("tridiag: frees");
    if (err != hipSuccess) { herm_free(c); char buf[256]; snprintf(buf, sizeof buf, "maus_herm_tridiag failed: %s", hipGetErrorString(err)); c->err = buf; return -1; }
    return 0;
}

int maus_herm_backtransform(maus_ctx* c, const double* z_real, int col_major) {
    const int n = c->rows;
    if (!c->hq || c->hqn != n || n != c->cols) FAIL(c, "maus_herm_backtransform: no reflectors for this matrix (maus_herm_tridiag first)");
    // z_real == NULL: the eigenvectors of T that maus_herm_tridiag_eig left on the device
    if (!z_real && (!c->hz || c->hzn != n)) FAIL(c, "maus_herm_backtransform: null input and no eigenvectors of T on the device (maus_herm_tridiag_eig)");
    if (z_real && c->hz) { (void)hipFree(c->hz); c->hz = nullptr; c->hzn = 0; }
    HermClock clk;
    const size_t nn = (size_t)n * n;
    if (c->vn != n) { if (c->V) (void)hipFree(c->V); c->V = nullptr; c->vn = 0; HIPCHK(c, hipMalloc((void**)&c->V, sizeof(c128) * nn)); c->vn = n; }
    double* Zr = nullptr; c128 *Vt = nullptr, *G = nullptr, *T = nullptr, *X = nullptr, *Y = nullptr;
    auto cleanup = [&]() { void* ps[] = {Zr, Vt, G, T, X, Y}; for (void* p : ps) if (p) (void)hipFree(p); };
#define HERM_ALLOC(ptr, bytes) do { if (hipMalloc((void**)&(ptr), (bytes)) != hipSuccess) { (void)hipGetLastError(); cleanup(); \
        FAIL(c, "maus_herm_backtransform: out of device memory"); } } while (0)
    if (z_real) HERM_ALLOC(Zr, sizeof(double) * nn);
    else { Zr = c->hz; c->hz = nullptr; c->hzn = 0; }          // ownership moves here: freed with the other temporaries
    HERM_ALLOC(Vt, sizeof(c128) * (size_t)HNB * n);
    HERM_ALLOC(G, sizeof(c128) * HNB * HNB);
    HERM_ALLOC(T, sizeof(c128) * HNB * HNB);
    HERM_ALLOC(X, sizeof(c128) * (size_t)HNB * n);
    HERM_ALLOC(Y, sizeof(c128) * (size_t)HNB * n);
#undef HERM_ALLOC
    clk.lap("backtransform: allocations");
    hipStream_t st = c->st;
    hipError_t err = hipSuccess;
    if (z_real && maus_stage_h2d(c, Zr, z_real, sizeof(double) * nn, st)) { cleanup(); return -1; }   // (the caller's Z is freed right after: see capi.hip on pinned staging)
    if (clk.on) { (void)hipStreamSynchronize(st); clk.lap("backtransform: upload of Z"); }
    if (err == hipSuccess) {
        if (col_major) hipLaunchKernelGGL(herm_transpose_to_complex_kernel, dim3((n + 31) / 32, (n + 31) / 32), dim3(256), 0, st, Zr, n, c->V);
        else hipLaunchKernelGGL(herm_real_to_complex_kernel, dim3((unsigned)((nn + HT - 1) / HT)), dim3(HT), 0, st, Zr, (long)nn, c->V);
    }
    const int nref = n - 1;                                   // reflectors H(0) .. H(n-2)
    const int nblocks = (nref + HNB - 1) / HNB;
    for (int b = nblocks - 1; b >= 0 && err == hipSuccess; --b) {
        const int k0 = b * HNB, nb = (nref - k0 < HNB) ? nref - k0 : HNB;
        const int r0 = k0 + 1, m = n - r0;                    // the block's reflectors are zero above row k0 + 1
        const c128* Vq = c->hq + (size_t)k0 * n + r0;         // [k][r], ld = n
        c128* Zs = c->V + (size_t)r0 * n;
        // G = V^H V (nb x nb), T from it (zlarft)
        maus_zgemm_launch(st, nb, nb, m, Vq, n, 0, Vq, n, 0, G, HNB, 0, 1.0, 0, 1, 1, true, false);
        hipLaunchKernelGGL(herm_larft_kernel, dim3(1), dim3(64), 0, st, G, nb, c->htau + k0, T);
        hipLaunchKernelGGL(herm_vt_kernel, dim3((unsigned)(((long)m * nb + HT - 1) / HT)), dim3(HT), 0, st, c->hq, n, k0, r0, nb, Vt);
        {   ProfScope ps(c, KC_GEMM, 8.0 * nb * (double)n * m * 2 + 8.0 * nb * (double)nb * n, 16.0 * ((double)m * n * 3));
            maus_zgemm_launch(st, nb, n, m, Vq, n, 0, Zs, n, 0, X, n, 0, 1.0, 0, 1, 0, true, false);         // X = V^H Z
            maus_zgemm_launch(st, nb, n, nb, T, HNB, 0, X, n, 0, Y, n, 0, 1.0, 0, 1, 0, false, false);       // Y = T X
            maus_zgemm_launch(st, m, n, nb, Vt, nb, 0, Y, n, 0, Zs, n, 0, -1.0, 1, 1, 0, false, false);      // Z -= V Y
        }
        err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipStreamSynchronize(st);
    if (err == hipSuccess) err = hipGetLastError();
    clk.lap("backtransform: kernels");
    cleanup();
    herm_free(c);                                            // the reflectors are only good for this one back-transformation
    clk.lap("backtransform: frees");
    if (err != hipSuccess) { char buf[256]; snprintf(buf, sizeof buf, "maus_herm_backtransform failed: %s", hipGetErrorString(err)); c->err = buf; return -1; }
    return 0;
}

}  // extern "C"

// Eigenvalues (ascending, to the host) and eigenvectors (left on the device for maus_herm_backtransform(ctx, NULL, 0)) of the
// real symmetric tridiagonal matrix (d, e).  diag_out[0] = smallest gap between neighbouring eigenvalues / ||T||,
// diag_out[1] = largest residual component max_k |(T z - lambda z)_k| over all vectors / ||T||, diag_out[2] = ||T|| (the
// Gershgorin radius); the caller decides from them whether to keep the result (see the kernels above).
static int tri_solve(maus_ctx* c, const double* d_host, const double* e_host, int n, double* w_out, double* diag_out, bool vectors) {
    if (!d_host || (!e_host && n > 1) || !w_out || (vectors && !diag_out) || n < 1) FAIL(c, "maus_herm_tridiag_eig: bad arguments");
    HermClock clk;
    if (c->hz) { (void)hipFree(c->hz); c->hz = nullptr; c->hzn = 0; }
    // Gershgorin interval and the scale of the problem (host: 2 n numbers)
    double gl = d_host[0], gu = d_host[0], emax = 0.0;
    for (int k = 0; k < n; ++k) {
        const double a = (k > 0 ? fabs(e_host[k - 1]) : 0.0) + (k + 1 < n ? fabs(e_host[k]) : 0.0);
        gl = std::min(gl, d_host[k] - a); gu = std::max(gu, d_host[k] + a);
        if (k + 1 < n) emax = std::max(emax, fabs(e_host[k]));
    }
    const double tnorm = std::max(fabs(gl), fabs(gu));
    if (!(tnorm < 1e150) || !(tnorm == tnorm)) FAIL(c, "maus_herm_tridiag_eig: the matrix is not finite or too large (scale it first)");
    const double eps = 2.220446049250313e-16, safmin = 2.2250738585072014e-308;
    const double pivmin = safmin * std::max(1.0, emax * emax);
    gl -= 2.0 * eps * n * tnorm + 2.0 * pivmin; gu += 2.0 * eps * n * tnorm + 2.0 * pivmin;
    std::vector<double> e2(std::max(1, n));
    for (int k = 0; k + 1 < n; ++k) e2[k] = e_host[k] * e_host[k];
    const size_t nn = (size_t)n * n;
    double *dd = nullptr, *de = nullptr, *de2 = nullptr, *dw = nullptr, *dn = nullptr, *dr = nullptr, *Z = nullptr, *B = nullptr;
    auto cleanup = [&]() { void* ps[] = {dd, de, de2, dw, dn, dr, Z, B}; for (void* p : ps) if (p) (void)hipFree(p); };
#define TRI_ALLOC(ptr, bytes) do { if (hipMalloc((void**)&(ptr), (bytes)) != hipSuccess) { (void)hipGetLastError(); cleanup(); \
        FAIL(c, "maus_herm_tridiag_eig: out of device memory"); } } while (0)
    TRI_ALLOC(dd, sizeof(double) * n); TRI_ALLOC(de, sizeof(double) * n); TRI_ALLOC(de2, sizeof(double) * n);
    TRI_ALLOC(dw, sizeof(double) * n); TRI_ALLOC(dn, sizeof(double) * n); TRI_ALLOC(dr, sizeof(double) * n);
    if (vectors) { TRI_ALLOC(Z, sizeof(double) * nn); TRI_ALLOC(B, sizeof(double) * nn); }
#undef TRI_ALLOC
    hipStream_t st = c->st;
    std::vector<double> ez(n, 0.0);
    for (int k = 0; k + 1 < n; ++k) ez[k] = e_host[k];
    if (maus_stage_h2d(c, dd, d_host, sizeof(double) * n, st) || maus_stage_h2d(c, de, ez.data(), sizeof(double) * n, st)
        || maus_stage_h2d(c, de2, e2.data(), sizeof(double) * n, st)) { cleanup(); return -1; }
    const dim3 g1((n + 255) / 256), b1(256);
    hipLaunchKernelGGL(tri_bisect_kernel, g1, b1, 0, st, dd, de2, n, gl, gu, pivmin, dw);
    if (clk.on) { (void)hipStreamSynchronize(st); clk.lap("tridiag eig: bisection"); }
    if (!vectors) {
        const int rc = maus_stage_d2h(c, w_out, dw, sizeof(double) * n, st);
        cleanup();
        return rc;
    }
    hipLaunchKernelGGL(tri_twisted_kernel, g1, b1, 0, st, dd, de, n, dw, eps * std::max(tnorm, safmin / eps), Z, B, dn);
    hipLaunchKernelGGL(tri_scale_kernel, dim3((n + 255) / 256, n), b1, 0, st, Z, n, dn);
    hipLaunchKernelGGL(tri_resid_kernel, g1, b1, 0, st, dd, de, n, dw, Z, dr);
    std::vector<double> res(n);
    hipError_t err = hipGetLastError();
    if (err == hipSuccess && (maus_stage_d2h(c, w_out, dw, sizeof(double) * n, st) || maus_stage_d2h(c, res.data(), dr, sizeof(double) * n, st))) { cleanup(); return -1; }
    if (err == hipSuccess) err = hipStreamSynchronize(st);
    if (err == hipSuccess) err = hipGetLastError();
    clk.lap("tridiag eig: vectors + checks");
    if (err != hipSuccess) { cleanup(); char buf[256]; snprintf(buf, sizeof buf, "maus_herm_tridiag_eig failed: %s", hipGetErrorString(err)); c->err = buf; return -1; }
    double gap = (n > 1) ? 1e300 : 1.0, rmax = 0.0;
    bool finite = true;
    for (int k = 0; k < n; ++k) {
        if (!(res[k] == res[k]) || !(w_out[k] == w_out[k])) finite = false;
        rmax = std::max(rmax, res[k]);
        if (k > 0) gap = std::min(gap, w_out[k] - w_out[k - 1]);
    }
    const double scale = tnorm > 0.0 ? tnorm : 1.0;
    diag_out[0] = (n > 1) ? gap / scale : 1.0;
    diag_out[1] = finite ? rmax / scale : 1e300;
    diag_out[2] = tnorm;
    c->hz = Z; c->hzn = n; Z = nullptr;                              // kept for the back-transformation
    cleanup();
    return 0;
}

int maus_herm_tridiag_eig(maus_ctx* c, const double* d, const double* e, int n, double* w_out, double* diag_out) {
    return tri_solve(c, d, e, n, w_out, diag_out, true);
}

// the eigenvalues alone (ascending): the bisection without the n x n work arrays -- singular values through the Hermitian
// embedding, the reporting prologue's spectra
int maus_herm_tridiag_eigvals(maus_ctx* c, const double* d, const double* e, int n, double* w_out) {
    return tri_solve(c, d, e, n, w_out, nullptr, false);
}
