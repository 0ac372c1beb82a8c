// Batched restarted GMRES with optional Jacobi (inverse-diagonal) left preconditioner (gfx950).
//
// Replaces, per candidate k, the reference's iterative branch (AMS:60-90):
//     spla.gmres(H_k, b, x0=b, tol=1e-8, maxiter=50, M=diag(1/diag(H_k)))   [tol -> rtol, SURVEY F2]
// whose algorithm is SciPy 1.15's scipy/sparse/linalg/_isolve/iterative.py:692-841: restart
// min(20, n), left preconditioning, modified Gram-Schmidt Arnoldi, Givens rotations in the
// LAPACK zlartg convention, adaptive inner tolerance `ptol`, true-residual exit
// ||b - H x|| <= rtol*||b||, info = maxiter when the restart cycles are exhausted.
//
// Shared-matrix mode (maus_gmres): H_k = A - shift_k I + psi_k I is never materialised: every inner iteration of ALL
// active candidates is ONE MFMA zgemm  Y = Z * A^T  (Z = the candidates' current Krylov vectors,
// gathered by row index) followed by a per-candidate kernel that adds (psi_k - shift_k) z,
// applies the Jacobi scale, orthogonalises (MGS, wavefront reductions), and runs the small
// Hessenberg / Givens / ptol state machine on one lane.  Each candidate advances through its
// own (cycle, column) state; the host only launches "ticks" until no candidate is active.
//
// Dense mode (maus_gmres_pert): the reference's H_solve of the GMRES branch also carries the random term
// 0.15 psi ((U1-.5) + i(U2-.5)) (AMS:49-52, 89).  At the default psi ~ 1e-19 that is below the rounding of one matvec
// and the shared-matrix mode is used; once psi has been escalated (attempt / stuck / aggression factors) it is not,
// and the candidates' H_k are materialised in the LU workspace (the same build kernels as the direct solve, i.e. the
// same bits as the reference's H_solve) -- the matvec is then one HBM-bound GEMV per candidate against its own H_k,
// and the Jacobi scale and its AMS:67-72 gate read diag(H_k) from the materialised matrix.
#include "ctx.h"
#include <algorithm>
#include <cstring>

void maus_zgemm_launch_idx(hipStream_t st, int M, int N, int K, const c128* A, long lda, long sA,
                           const c128* B, long ldb, long sB, c128* C, long ldc, long sC,
                           double alpha, int beta, int batch, int blay, bool conja, bool conjb,
                           const int* a_rows, const int* c_rows);

namespace {

constexpr int MAXR = 20;
constexpr int GT = 256;
constexpr double EPS = 2.220446049250313e-16;

struct GState {
    int phase;        // 1: residual pending (z = x); 0: Arnoldi step pending (z = V[col]); 2: done
    int col, cycle, inner, info, breakdown, first, pad;
    double bnrm2, atol, ptol, pmf, presid, rnorm;
    double gc[MAXR];
    c128 gs[MAXR];
    c128 S[MAXR + 1];
    c128 h[MAXR][MAXR + 1];
};

__device__ __forceinline__ double blk_sum(double v, double* sbuf) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sbuf[wave] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < GT / 64; ++w) t += sbuf[w];
    return t;
}

// NumPy's complex reciprocal 1.0 / z  (Smith, as (1+0j)/z)
__device__ __forceinline__ c128 crecip_np(c128 z) {
    if (fabs(z.x) >= fabs(z.y)) {
        if (z.x == 0.0 && z.y == 0.0) return cmake(1.0 / fabs(z.x), 0.0 / fabs(z.y));   // inf, nan like NumPy
        const double rat = z.y / z.x, scl = 1.0 / (z.x + z.y * rat);
        return cmake(scl, -rat * scl);
    }
    const double rat = z.x / z.y, scl = 1.0 / (z.y + z.x * rat);
    return cmake(rat * scl, -scl);
}
__device__ __forceinline__ c128 cdiv_np(c128 a, c128 b) {
    if (fabs(b.x) >= fabs(b.y)) {
        const double rat = b.y / b.x, scl = 1.0 / (b.x + b.y * rat);
        return cmake((a.x + a.y * rat) * scl, (a.y - a.x * rat) * scl);
    }
    const double rat = b.x / b.y, scl = 1.0 / (b.y + b.x * rat);
    return cmake((a.x * rat + a.y) * scl, (a.y * rat - a.x) * scl);
}

// LAPACK 3.10+ zlartg, safe-range branches:  [c s; -conj(s) c] [f; g] = [r; 0], c real
__device__ void zlartg_dev(c128 f, c128 g, double& c, c128& s, c128& r) {
    const double safmin = 2.2250738585072014e-308, rtmin = 1.4916681462400413e-154;
    if (g.x == 0.0 && g.y == 0.0) { c = 1.0; s = cmake(0.0, 0.0); r = f; return; }
    if (f.x == 0.0 && f.y == 0.0) {
        c = 0.0;
        double d;
        if (g.x == 0.0) d = fabs(g.y); else if (g.y == 0.0) d = fabs(g.x); else d = sqrt(g.x * g.x + g.y * g.y);
        s = cmake(g.x / d, -g.y / d); r = cmake(d, 0.0); return;
    }
    const double f2 = f.x * f.x + f.y * f.y, g2 = g.x * g.x + g.y * g.y, h2 = f2 + g2;
    const c128 gc_ = cconj(g);
    if (f2 >= h2 * safmin) {
        c = sqrt(f2 / h2);
        r = cmake(f.x / c, f.y / c);
        const double rtmax2 = 6.703903964971299e+153;    // 2 * sqrt(safmax/4)
        if (f2 > rtmin && h2 < rtmax2) { const double d = sqrt(f2 * h2); s = cmul(gc_, cmake(f.x / d, f.y / d)); }
        else s = cmul(gc_, cmake(r.x / h2, r.y / h2));
    } else {
        const double d = sqrt(f2 * h2);
        c = f2 / d;
        if (c >= safmin) r = cmake(f.x / c, f.y / c); else { const double q = h2 / d; r = cmake(f.x * q, f.y * q); }
        s = cmul(gc_, cmake(f.x / d, f.y / d));
    }
}

__global__ void __launch_bounds__(256)
diag_kernel(const c128* __restrict__ A, int n, c128* __restrict__ d) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = A[(long)i * n + i];
}

// AMS:67-72: ok iff every 1/diag(H) is finite and every |diag(H)| > 1e-12
__global__ void __launch_bounds__(GT)
jacobi_check_kernel(const c128* __restrict__ dA, int n, const c128* __restrict__ shift, const double* __restrict__ psi,
                    int* __restrict__ ok) {
    __shared__ int sbad;
    const int g = blockIdx.x;
    if (threadIdx.x == 0) sbad = 0;
    __syncthreads();
    const c128 lam = shift[g]; const double ps = psi[g];
    bool bad = false;
    for (int i = threadIdx.x; i < n; i += GT) {
        const c128 d = cmake(__dadd_rn(__dsub_rn(dA[i].x, lam.x), ps), __dadd_rn(__dsub_rn(dA[i].y, lam.y), 0.0));
        const c128 inv = crecip_np(d);
        if (!cfinite(inv) || !(hypot(d.x, d.y) > 1e-12)) bad = true;
    }
    if (bad) atomicOr(&sbad, 1);
    __syncthreads();
    if (threadIdx.x == 0) ok[g] = sbad ? 0 : 1;
}

struct GArgs {
    int n, R, maxiter, rows_per;          // rows_per = R + 2 rows of the basis array per candidate
    double rtol;
    const c128* dA; const c128* shift; const double* psi; const int* jac;
    const c128* X; long ldx; const int* slots; const c128* bvec; int rhs_mode;
    c128* Vb; c128* Y; GState* st;
    const c128* Hd; long ldh, strideH;    // dense mode: materialised H_k (null in shared-matrix mode)
};

__device__ __forceinline__ const c128* rhs_of(const GArgs& a, int k) {
    return a.rhs_mode == 0 ? a.X + (long)a.slots[k] * a.ldx : a.bvec;
}
__device__ __forceinline__ c128 psolve1(const GArgs& a, int k, int i, c128 v) {
    if (!a.jac[k]) return v;
    c128 d;
    if (a.Hd) d = a.Hd[(long)k * a.strideH + (long)i * a.ldh + i];
    else {
        const c128 lam = a.shift[k];
        d = cmake(__dadd_rn(__dsub_rn(a.dA[i].x, lam.x), a.psi[k]), __dadd_rn(__dsub_rn(a.dA[i].y, lam.y), 0.0));
    }
    return cmul(crecip_np(d), v);
}

// dense mode, AMS:65-86: Jacobi only where it was asked for AND every 1/diag(H_k) is finite and every |diag(H_k)| > 1e-12
__global__ void __launch_bounds__(GT)
jacobi_gate_dense_kernel(GArgs a, int* __restrict__ jac) {
    __shared__ int sbad;
    const int k = blockIdx.x;
    if (threadIdx.x == 0) sbad = 0;
    __syncthreads();
    bool bad = false;
    if (jac[k]) {
        for (int i = threadIdx.x; i < a.n; i += GT) {
            const c128 d = a.Hd[(long)k * a.strideH + (long)i * a.ldh + i];
            const c128 inv = crecip_np(d);
            if (!cfinite(inv) || !(hypot(d.x, d.y) > 1e-12)) bad = true;
        }
    }
    if (bad) atomicOr(&sbad, 1);
    __syncthreads();
    if (threadIdx.x == 0 && sbad) jac[k] = 0;
}

// dense mode matvec: Y[k] = H_k z_k for the active candidates.  One wave per matrix row at a time (a row is n x 16
// contiguous bytes), 16 rows per workgroup, z through L2: HBM-bound on H_k.
__global__ void __launch_bounds__(GT)
gemv_dense_kernel(GArgs a, const int* __restrict__ act, const int* __restrict__ zrow) {
    const int k = act[blockIdx.y];
    const c128* H = a.Hd + (long)k * a.strideH;
    const c128* z = a.Vb + (long)zrow[blockIdx.y] * a.n;
    c128* y = a.Y + (long)k * a.n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * 16 + wave * 4;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int i = r0 + rr;
        if (i >= a.n) break;
        const c128* row = H + (long)i * a.ldh;
        double sr = 0.0, si = 0.0;
        for (int j = lane; j < a.n; j += 64) {
            const c128 h = row[j], v = z[j];
            sr = fma(h.x, v.x, sr); sr = fma(-h.y, v.y, sr);
            si = fma(h.x, v.y, si); si = fma(h.y, v.x, si);
        }
        sr = wave_sum(sr); si = wave_sum(si);
        if (lane == 0) y[i] = cmake(sr, si);
    }
}

// x0 = b; norms; ptol (iterative.py:696-724)
__global__ void __launch_bounds__(GT)
gmres_init_kernel(GArgs a) {
    __shared__ double sbuf[GT / 64];
    const int k = blockIdx.x;
    const c128* b = rhs_of(a, k);
    c128* x = a.Vb + ((long)k * a.rows_per + a.R + 1) * a.n;
    double sb = 0.0, sm = 0.0;
    for (int i = threadIdx.x; i < a.n; i += GT) {
        const c128 v = b[i];
        x[i] = v;
        sb = fma(v.x, v.x, sb); sb = fma(v.y, v.y, sb);
        const c128 m = psolve1(a, k, i, v);
        sm = fma(m.x, m.x, sm); sm = fma(m.y, m.y, sm);
    }
    sb = blk_sum(sb, sbuf); sm = blk_sum(sm, sbuf);
    if (threadIdx.x == 0) {
        GState& s = a.st[k];
        s.bnrm2 = sqrt(sb);
        s.atol = fmax(0.0, a.rtol * s.bnrm2);
        s.pmf = 1.0;
        s.ptol = sqrt(sm) * fmin(1.0, s.atol / s.bnrm2);
        s.presid = 0.0; s.rnorm = 0.0;
        s.col = 0; s.cycle = 0; s.inner = 0; s.info = 0; s.breakdown = 0; s.first = 1;
        s.phase = (s.bnrm2 == 0.0) ? 2 : 1;             // b == 0 -> return b, info 0
    }
}

// Y[act[first + p]] = Y[act[0]] (see the first product of a linear system in maus_gmres_run)
__global__ void __launch_bounds__(256)
bcast_row_kernel(c128* __restrict__ Y, int n, const int* __restrict__ act, int first) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) Y[(long)act[first + blockIdx.y] * n + i] = Y[(long)act[0] * n + i];
}

// active list + the row of the basis array each active candidate multiplies next
__global__ void __launch_bounds__(1024)
gmres_compact_kernel(const GState* __restrict__ st, int count, int rows_per, int R, int* __restrict__ act,
                     int* __restrict__ zrow, int* __restrict__ nact) {
    __shared__ int scnt[16];
    __shared__ int sbase;
    if (threadIdx.x == 0) sbase = 0;
    __syncthreads();
    for (int k0 = 0; k0 < count; k0 += 1024) {
        const int k = k0 + threadIdx.x;
        const bool on = k < count && st[k].phase != 2;
        const unsigned long long m = __ballot(on);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) scnt[wave] = __popcll(m);
        __syncthreads();
        int off = sbase;
        for (int w = 0; w < wave; ++w) off += scnt[w];
        if (on) {
            const int p = off + __popcll(m & ((1ull << lane) - 1ull));
            act[p] = k;
            zrow[p] = k * rows_per + (st[k].phase == 1 ? R + 1 : st[k].col);
        }
        __syncthreads();
        if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += scnt[w]; sbase += t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *nact = sbase;
}

template <int EPT>
__global__ void __launch_bounds__(GT)
gmres_post_kernel(GArgs a, const int* __restrict__ act) {
    __shared__ double sbuf[GT / 64];
    __shared__ c128 s_h[MAXR + 2];
    __shared__ c128 s_y[MAXR + 1];
    __shared__ int s_flag[2];                 // [0]: x update requested, [1]: candidate finished this tick
    const int k = act[blockIdx.x];
    GState& s = a.st[k];
    const int n = a.n, R = a.R, tid = threadIdx.x;
    const int phase = s.phase, col = s.col;
    c128* Vk = a.Vb + (long)k * a.rows_per * n;
    c128* x = Vk + (long)(R + 1) * n;
    const c128* y = a.Y + (long)k * n;
    const c128 lam = a.shift[k];
    // H z = A z + (psi - lambda) z ; in dense mode the shift is part of the materialised H_k
    const c128 sh = a.Hd ? cmake(0.0, 0.0) : cmake(a.psi[k] - lam.x, -lam.y);
    c128 w[EPT];

    if (phase == 1) {
        // ---- r = b - H x ; exit tests ; start of the next restart cycle (iterative.py:737-748, 826-838) ----
        const c128* b = rhs_of(a, k);
        double ss = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + e * GT;
            if (i < n) {
                c128 hx = y[i]; cfma(hx, sh, x[i]);
                const c128 r = cmake(b[i].x - hx.x, b[i].y - hx.y);
                w[e] = r;
                ss = fma(r.x, r.x, ss); ss = fma(r.y, r.y, ss);
            }
        }
        const double rnorm = sqrt(blk_sum(ss, sbuf));
        bool done = false; int info = 0;
        double pmf = s.pmf, ptol = s.ptol; int cycle = s.cycle;
        if (!(rnorm == rnorm)) { done = true; info = a.maxiter; }              // NaN: can never pass a test
        else if (s.first) { if (rnorm < s.atol) { done = true; info = 0; } }
        else {
            if (rnorm <= s.atol) { done = true; info = 0; }
            else if (s.breakdown) { done = true; info = a.maxiter; }
            else {
                if (s.presid <= s.ptol) pmf = fmax(EPS, 0.25 * pmf); else pmf = fmin(1.0, 1.5 * pmf);
                ptol = s.presid * fmin(pmf, s.atol / rnorm);
                cycle += 1;
                if (cycle >= a.maxiter) { done = true; info = a.maxiter; }
            }
        }
        if (done) {
            __syncthreads();
            if (tid == 0) { s.rnorm = rnorm; s.info = info; s.phase = 2; }
            return;
        }
        // V[0] = psolve(r) / ||psolve(r)||
        double sv = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + e * GT;
            if (i < n) { w[e] = psolve1(a, k, i, w[e]); sv = fma(w[e].x, w[e].x, sv); sv = fma(w[e].y, w[e].y, sv); }
        }
        const double tmp = sqrt(blk_sum(sv, sbuf));
        const double inv = 1.0 / tmp;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + e * GT;
            if (i < n) Vk[i] = cmake(w[e].x * inv, w[e].y * inv);
        }
        if (tid == 0) {
            s.rnorm = rnorm; s.pmf = pmf; s.ptol = ptol; s.cycle = cycle; s.first = 0;
            for (int j = 0; j <= R; ++j) s.S[j] = cmake(0.0, 0.0);
            s.S[0] = cmake(tmp, 0.0);
            s.col = 0; s.breakdown = 0; s.phase = 0;
        }
        return;
    }

    // ---- Arnoldi step (iterative.py:751-800) ----
    const c128* z = Vk + (long)col * n;
    double ss = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * GT;
        if (i < n) {
            c128 av = y[i]; cfma(av, sh, z[i]);
            w[e] = psolve1(a, k, i, av);
            ss = fma(w[e].x, w[e].x, ss); ss = fma(w[e].y, w[e].y, ss);
        }
    }
    const double h0 = sqrt(blk_sum(ss, sbuf));
    for (int kk = 0; kk <= col; ++kk) {                        // modified Gram-Schmidt
        const c128* vk = Vk + (long)kk * n;
        double tr = 0.0, ti = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + e * GT;
            if (i < n) {
                const c128 v = vk[i];
                tr = fma(v.x, w[e].x, tr); tr = fma(v.y, w[e].y, tr);
                ti = fma(v.x, w[e].y, ti); ti = fma(-v.y, w[e].x, ti);
            }
        }
        tr = blk_sum(tr, sbuf); ti = blk_sum(ti, sbuf);
        const c128 t = cmake(tr, ti);
        if (tid == 0) s_h[kk] = t;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + e * GT;
            if (i < n) cfms(w[e], t, vk[i]);
        }
    }
    ss = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * GT;
        if (i < n) { ss = fma(w[e].x, w[e].x, ss); ss = fma(w[e].y, w[e].y, ss); }
    }
    const double h1 = sqrt(blk_sum(ss, sbuf));
    const bool brk = h1 <= EPS * h0;
    {
        c128* vn = Vk + (long)(col + 1) * n;
        const double inv = brk ? 1.0 : 1.0 / h1;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + e * GT;
            if (i < n) vn[i] = cmake(w[e].x * inv, w[e].y * inv);
        }
    }
    if (tid == 0) {
        // Hessenberg column: previous rotations, new rotation, rhs S, ptol test
        s_h[col + 1] = cmake(brk ? 0.0 : h1, 0.0);
        for (int kk = 0; kk < col; ++kk) {
            const double c = s.gc[kk]; const c128 sg = s.gs[kk];
            const c128 n0 = s_h[kk], n1 = s_h[kk + 1];
            c128 a0 = cmake(c * n0.x, c * n0.y); cfma(a0, sg, n1);
            c128 a1 = cmake(c * n1.x, c * n1.y); cfms(a1, cconj(sg), n0);
            s_h[kk] = a0; s_h[kk + 1] = a1;
        }
        double c; c128 sg, mag;
        zlartg_dev(s_h[col], s_h[col + 1], c, sg, mag);
        s.gc[col] = c; s.gs[col] = sg;
        s_h[col] = mag; s_h[col + 1] = cmake(0.0, 0.0);
        const c128 Sc = s.S[col];
        const c128 tmp = cmul(cmake(-sg.x, sg.y), Sc);          // -conj(s) * S[col]
        s.S[col] = cmake(c * Sc.x, c * Sc.y);
        s.S[col + 1] = tmp;
        const double presid = hypot(tmp.x, tmp.y);
        s.presid = presid;
        s.inner += 1;
        for (int j = 0; j <= col + 1; ++j) s.h[col][j] = s_h[j];
        if (brk) s.breakdown = 1;
        const bool end_inner = (presid <= s.ptol) || brk || (col == R - 1);
        s_flag[0] = end_inner ? 1 : 0;
        if (end_inner) {
            // y = trsv(h^T, S) with the pseudo-solve rules of iterative.py:806-816
            if (s.h[col][col].x == 0.0 && s.h[col][col].y == 0.0) s.S[col] = cmake(0.0, 0.0);
            for (int j = 0; j <= col; ++j) s_y[j] = s.S[j];
            for (int kk = col; kk > 0; --kk) {
                if (s_y[kk].x != 0.0 || s_y[kk].y != 0.0) {
                    s_y[kk] = cdiv_np(s_y[kk], s.h[kk][kk]);
                    const c128 t = s_y[kk];
                    for (int j = 0; j < kk; ++j) cfms(s_y[j], t, s.h[kk][j]);
                }
            }
            if (s_y[0].x != 0.0 || s_y[0].y != 0.0) s_y[0] = cdiv_np(s_y[0], s.h[0][0]);
            s.phase = 1;
        } else {
            s.col = col + 1;
        }
    }
    __syncthreads();
    if (s_flag[0]) {                                           // x += y @ V[:col+1]
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + e * GT;
            if (i < n) {
                c128 acc = x[i];
                for (int j = 0; j <= col; ++j) cfma(acc, s_y[j], Vk[(long)j * n + i]);
                x[i] = acc;
            }
        }
    }
}

// W[slot] <- x ; outputs
__global__ void __launch_bounds__(GT)
gmres_finish_kernel(GArgs a, c128* __restrict__ W, long ldw, int* __restrict__ info, int* __restrict__ inner, int* __restrict__ status) {
    __shared__ int sbad;
    const int k = blockIdx.x;
    if (threadIdx.x == 0) sbad = 0;
    __syncthreads();
    const c128* x = a.Vb + ((long)k * a.rows_per + a.R + 1) * a.n;
    c128* w = W + (long)a.slots[k] * ldw;
    bool bad = false;
    for (int i = threadIdx.x; i < a.n; i += GT) { const c128 v = x[i]; w[i] = v; bad |= !cfinite(v); }
    if (bad) atomicOr(&sbad, 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        const GState& s = a.st[k];
        info[k] = (s.phase == 2) ? s.info : a.maxiter;
        inner[k] = s.inner;
        status[k] = (info[k] == 0 && sbad) ? -2 : 0;
    }
}

}  // namespace

int maus_jacobi_check_run(maus_ctx* c, int count, const double* shift, const double* psi, int32_t* ok) {
    if (!c->A || c->rows != c->cols) FAIL(c, "maus_jacobi_check: square matrix required");
    if (count <= 0) return 0;
    const int n = c->rows;
    if (ensure_scalars(c, count)) return -1;
    if (ensure_scratch(c, sizeof(c128) * n)) return -1;
    c128* dA = (c128*)c->scratch;
    hipLaunchKernelGGL(diag_kernel, dim3((n + 255) / 256), dim3(256), 0, c->st, c->A, n, dA);
    if (maus_h2d(c, c->d_c1, shift, sizeof(c128) * count, c->st)) return -1;
    if (maus_h2d(c, c->d_r1, psi, sizeof(double) * count, c->st)) return -1;
    hipLaunchKernelGGL(jacobi_check_kernel, dim3(count), dim3(GT), 0, c->st, dA, n, c->d_c1, c->d_r1, c->d_i1);
    if (maus_d2h(c, ok, c->d_i1, sizeof(int) * count, c->st)) return -1;
    HIPCHK(c, hipStreamSynchronize(c->st));
    return 0;
}

int maus_gmres_run(maus_ctx* c, const int* slots, int count, const double* shift, const double* psi, int rhs_mode,
                   const int32_t* use_jacobi, double rtol, int restart, int maxiter, int32_t* info_out, int32_t* inner_out,
                   int32_t* status, const c128* Hdense, long ldh, long strideH, int32_t* jacobi_out) {
    if (!c->A || !c->X) FAIL(c, "maus_gmres: matrix/population missing");
    if (c->rows != c->cols) FAIL(c, "maus_gmres: square matrix required");
    if (rhs_mode == 1 && (!c->b || c->bn != c->rows)) FAIL(c, "maus_gmres: rhs b not set");
    if (count <= 0) return 0;
    const int n = c->rows;
    if (n > 32 * GT) FAIL(c, "maus_gmres: n <= 8192 in this build");
    const int R = std::max(1, std::min(std::min(restart, MAXR), n));
    if (maxiter < 1) maxiter = 1;
    if (upload_slots(c, slots, count)) return -1;
    // scratch: diag | basis (count*(R+2) rows) | Y (count rows) | states | act | zrow | nact | jac | info/inner/status
    const size_t rows_per = (size_t)R + 2;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t o_d = take(sizeof(c128) * n), o_v = take(sizeof(c128) * rows_per * count * n), o_y = take(sizeof(c128) * (size_t)count * n),
                 o_s = take(sizeof(GState) * count), o_a = take(sizeof(int) * count), o_z = take(sizeof(int) * count),
                 o_n = take(sizeof(int)), o_j = take(sizeof(int) * count), o_o = take(sizeof(int) * 3 * count);
    if (ensure_scratch(c, off)) return -1;
    char* base = (char*)c->scratch;
    GArgs a;
    a.n = n; a.R = R; a.maxiter = maxiter; a.rows_per = (int)rows_per; a.rtol = rtol;
    a.dA = (c128*)(base + o_d); a.Vb = (c128*)(base + o_v); a.Y = (c128*)(base + o_y); a.st = (GState*)(base + o_s);
    int* act = (int*)(base + o_a); int* zrow = (int*)(base + o_z); int* nact = (int*)(base + o_n);
    int* jac = (int*)(base + o_j); int* outs = (int*)(base + o_o);
    a.Hd = Hdense; a.ldh = ldh; a.strideH = strideH;
    a.shift = c->d_c1; a.psi = c->d_r1; a.jac = jac; a.X = c->X; a.ldx = c->ldp; a.slots = c->d_slots; a.bvec = c->b; a.rhs_mode = rhs_mode;
    if (maus_h2d(c, c->d_c1, shift, sizeof(c128) * count, c->st)) return -1;
    if (maus_h2d(c, c->d_r1, psi, sizeof(double) * count, c->st)) return -1;
    if (maus_h2d(c, jac, use_jacobi, sizeof(int) * count, c->st)) return -1;
    hipLaunchKernelGGL(diag_kernel, dim3((n + 255) / 256), dim3(256), 0, c->st, c->A, n, (c128*)(base + o_d));
    if (Hdense) hipLaunchKernelGGL(jacobi_gate_dense_kernel, dim3(count), dim3(GT), 0, c->st, a, jac);
    hipLaunchKernelGGL(gmres_init_kernel, dim3(count), dim3(GT), 0, c->st, a);
    const long max_ticks = (long)maxiter * (R + 1) + 2;
    int h_nact = 0;
    for (long tick = 0; tick < max_ticks; ++tick) {
        hipLaunchKernelGGL(gmres_compact_kernel, dim3(1), dim3(1024), 0, c->st, a.st, count, (int)rows_per, R, act, zrow, nact);
        if (maus_d2h(c, &h_nact, nact, sizeof(int), c->st)) return -1;
        HIPCHK(c, hipStreamSynchronize(c->st));
        if (h_nact <= 0) break;
        if (Hdense) {
            ProfScope ps(c, KC_VEC, 0, 16.0 * h_nact * (double)n * n);
            hipLaunchKernelGGL(gemv_dense_kernel, dim3((n + 15) / 16, h_nact), dim3(GT), 0, c->st, a, act, zrow);
        } else
        if (tick == 0 && rhs_mode == 1 && h_nact > 33) {
            // The first product of a linear system is A x0 with x0 = b for EVERY candidate (AMS:85: x0 = b; the shift and psi come in
            // behind the product): 33 rows of it -- enough for the kernel family of the full product, so the same bits (round 4:
            // tests/test_gpu_kernels.py, rows do not depend on the batch) -- and a copy for the others, instead of the same
            // row of a 512-row product 512 times.
            { ProfScope ps(c, KC_GEMM, 8.0 * 33 * (double)n * n, 16.0 * ((double)n * n + 2.0 * 33 * n));
              maus_zgemm_launch_idx(c->st, 33, n, n, a.Vb, n, 0, c->A, n, 0, a.Y, n, 0, 1.0, 0, 1, 1, false, false, zrow, act); }
            hipLaunchKernelGGL(bcast_row_kernel, dim3((n + 255) / 256, h_nact - 33), dim3(256), 0, c->st, a.Y, n, act, 33);
        } else
        { ProfScope ps(c, KC_GEMM, 8.0 * h_nact * (double)n * n, 16.0 * ((double)n * n + 2.0 * h_nact * n));
          maus_zgemm_launch_idx(c->st, h_nact, n, n, a.Vb, n, 0, c->A, n, 0, a.Y, n, 0, 1.0, 0, 1, 1, false, false, zrow, act); }
        { ProfScope ps(c, KC_VEC, 0, 16.0 * h_nact * (double)n * (R + 4));
          if (n <= 4 * GT) hipLaunchKernelGGL((gmres_post_kernel<4>), dim3(h_nact), dim3(GT), 0, c->st, a, act);
          else if (n <= 16 * GT) hipLaunchKernelGGL((gmres_post_kernel<16>), dim3(h_nact), dim3(GT), 0, c->st, a, act);
          else hipLaunchKernelGGL((gmres_post_kernel<32>), dim3(h_nact), dim3(GT), 0, c->st, a, act); }
    }
    hipLaunchKernelGGL(gmres_finish_kernel, dim3(count), dim3(GT), 0, c->st, a, c->W, c->ldp, outs, outs + count, outs + 2 * count);
    if (maus_d2h(c, info_out, outs, sizeof(int) * count, c->st)) return -1;
    if (maus_d2h(c, inner_out, outs + count, sizeof(int) * count, c->st)) return -1;
    if (maus_d2h(c, status, outs + 2 * count, sizeof(int) * count, c->st)) return -1;
    if (jacobi_out) { if (maus_d2h(c, jacobi_out, jac, sizeof(int) * count, c->st)) return -1; }
    HIPCHK(c, hipStreamSynchronize(c->st));
    HIPCHK(c, hipGetLastError());
    return 0;
}
