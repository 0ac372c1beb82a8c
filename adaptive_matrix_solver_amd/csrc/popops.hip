// Per-candidate vector kernels of the MAUS step (gfx950): HBM-bound streaming passes over the
// population rows with wavefront reductions for dot / norm.  One workgroup per candidate.
//   rayleigh dots        AMS:264-268      relax + normalise   AMS:280-283 / 285
//   residual norms       AMS:295-301      finite scan         AMS:319-327
//   SVD normalisations   AMS:233-242      Hermitian arg-max   AMS:165-173
#include "common.h"

namespace {

constexpr int VT = 256;   // threads per candidate

__device__ __forceinline__ double block_sum(double v, double* sbuf) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sbuf[wave] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < VT / 64; ++w) t += sbuf[w];
    return t;
}

// num = sum conj(x) * y ; den = sum conj(x) * x          (np.vdot conjugates its first argument)
__global__ void __launch_bounds__(VT)
rayleigh_dots_kernel(const c128* __restrict__ X, const c128* __restrict__ Y, long ld, const int* __restrict__ slots,
                     int n, c128* __restrict__ num, c128* __restrict__ den)
{
    __shared__ double sbuf[VT / 64];
    const int g = blockIdx.x;
    const c128* x = X + (long)slots[g] * ld;
    const c128* y = Y + (long)slots[g] * ld;
    double nr = 0, ni = 0, dr = 0;
    for (int i = threadIdx.x; i < n; i += VT) {
        const c128 a = x[i], b = y[i];
        nr = fma(a.x, b.x, nr); nr = fma(a.y, b.y, nr);
        ni = fma(a.x, b.y, ni); ni = fma(-a.y, b.x, ni);
        dr = fma(a.x, a.x, dr); dr = fma(a.y, a.y, dr);
    }
    nr = block_sum(nr, sbuf); ni = block_sum(ni, sbuf); dr = block_sum(dr, sbuf);
    if (threadIdx.x == 0) { num[g] = cmake(nr, ni); den[g] = cmake(dr, 0.0); }
}

// x <- (1-alpha) x + alpha w ; nrm = ||x|| ; if (normalise && nrm > 1e-10) x *= 1/nrm
// The vector stays in registers between the two passes (up to 16384 entries per candidate,
// longer vectors take the re-read path).
__global__ void __launch_bounds__(VT)
relax_normalise_kernel(c128* __restrict__ X, const c128* __restrict__ W, long ld, const int* __restrict__ slots,
                       int n, const c128* __restrict__ alpha, int normalise, double* __restrict__ norm_out)
{
    __shared__ double sbuf[VT / 64];
    const int g = blockIdx.x;
    c128* x = X + (long)slots[g] * ld;
    const c128* w = W + (long)slots[g] * ld;
    const c128 al = alpha[g];
    const c128 om = cmake(__dsub_rn(1.0, al.x), -al.y);
    double ss = 0.0;
    for (int i = threadIdx.x; i < n; i += VT) {
        const c128 a = x[i], b = w[i];
        // NumPy rounding order: each complex product rounded, then the sum
        double t1r = __dsub_rn(__dmul_rn(om.x, a.x), __dmul_rn(om.y, a.y));
        double t1i = __dadd_rn(__dmul_rn(om.x, a.y), __dmul_rn(om.y, a.x));
        double t2r = __dsub_rn(__dmul_rn(al.x, b.x), __dmul_rn(al.y, b.y));
        double t2i = __dadd_rn(__dmul_rn(al.x, b.y), __dmul_rn(al.y, b.x));
        c128 v = cmake(__dadd_rn(t1r, t2r), __dadd_rn(t1i, t2i));
        x[i] = v;
        ss = fma(v.x, v.x, ss); ss = fma(v.y, v.y, ss);
    }
    ss = block_sum(ss, sbuf);
    const double nrm = sqrt(ss);
    if (threadIdx.x == 0) norm_out[g] = nrm;
    if (normalise && nrm > 1e-10) {
        const double inv = 1.0 / nrm;          // NumPy divides complex by real via the reciprocal
        for (int i = threadIdx.x; i < n; i += VT) {     // own elements: program order suffices
            c128 v = x[i];
            x[i] = cmake(__dmul_rn(v.x, inv), __dmul_rn(v.y, inv));
        }
    }
}

// kind 1: r = ||y - lam*x|| ; kind 2: r = ||y - b|| ; also finite scan of x
__global__ void __launch_bounds__(VT)
residual_kernel(int kind, const c128* __restrict__ X, const c128* __restrict__ Y, long ld, const int* __restrict__ slots,
                int n, const c128* __restrict__ lam, const c128* __restrict__ bvec,
                double* __restrict__ resid, int* __restrict__ finite)
{
    __shared__ double sbuf[VT / 64];
    __shared__ int sbad;
    const int g = blockIdx.x;
    if (threadIdx.x == 0) sbad = 0;
    const c128* x = X + (long)slots[g] * ld;
    const c128* y = Y + (long)slots[g] * ld;
    const c128 l = lam ? lam[g] : cmake(0.0, 0.0);
    double ss = 0.0; bool bad = false;
    for (int i = threadIdx.x; i < n; i += VT) {
        const c128 a = x[i], b = y[i];
        c128 d;
        if (kind == 1) {
            double tr = __dsub_rn(__dmul_rn(l.x, a.x), __dmul_rn(l.y, a.y));
            double ti = __dadd_rn(__dmul_rn(l.x, a.y), __dmul_rn(l.y, a.x));
            d = cmake(__dsub_rn(b.x, tr), __dsub_rn(b.y, ti));
        } else {
            const c128 bb = bvec[i];
            d = cmake(__dsub_rn(b.x, bb.x), __dsub_rn(b.y, bb.y));
        }
        ss = fma(d.x, d.x, ss); ss = fma(d.y, d.y, ss);
        bad |= !cfinite(a);
    }
    ss = block_sum(ss, sbuf);
    if (bad) atomicOr(&sbad, 1);
    __syncthreads();
    if (threadIdx.x == 0) { resid[g] = sqrt(ss); finite[g] = sbad ? 0 : 1; }
}

// SVD residual pieces: r = ||y - sigma*u||  (y = A v, or A^H u against v)
__global__ void __launch_bounds__(VT)
svd_resid_kernel(const c128* __restrict__ Yv, const c128* __restrict__ Uv, long ld, const int* __restrict__ slots,
                 int n, const c128* __restrict__ sigma, double* __restrict__ out, int accumulate,
                 int* __restrict__ finite)
{
    __shared__ double sbuf[VT / 64];
    __shared__ int sbad;
    const int g = blockIdx.x;
    if (threadIdx.x == 0) sbad = 0;
    const c128* y = Yv + (long)slots[g] * ld;
    const c128* u = Uv + (long)slots[g] * ld;
    const double s = sigma[g].x;
    double ss = 0.0; bool bad = false;
    for (int i = threadIdx.x; i < n; i += VT) {
        const c128 a = y[i], b = u[i];
        const double dr = __dsub_rn(a.x, __dmul_rn(s, b.x)), di = __dsub_rn(a.y, __dmul_rn(s, b.y));
        ss = fma(dr, dr, ss); ss = fma(di, di, ss);
        bad |= !cfinite(b);
    }
    ss = block_sum(ss, sbuf);
    if (bad) atomicOr(&sbad, 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        const double r = sqrt(ss);
        out[g] = accumulate ? out[g] + r : r;
        if (accumulate) finite[g] = (finite[g] && !sbad) ? 1 : 0; else finite[g] = sbad ? 0 : 1;
    }
}

// nrm = ||src|| ; dst = src * (1 / (nrm > 1e-10 ? nrm : 1))     (AMS:234-235, 241-242)
__global__ void __launch_bounds__(VT)
norm_scale_kernel(const c128* S, c128* D, long ld, const int* __restrict__ slots, int n,      // S == D is a legitimate call (in place): no __restrict__ on them
                  double* __restrict__ norm_out, int stride_out, int off_out)
{
    __shared__ double sbuf[VT / 64];
    const int g = blockIdx.x;
    const c128* s = S + (long)slots[g] * ld;
    c128* d = D + (long)slots[g] * ld;
    double ss = 0.0;
    for (int i = threadIdx.x; i < n; i += VT) { const c128 a = s[i]; ss = fma(a.x, a.x, ss); ss = fma(a.y, a.y, ss); }
    ss = block_sum(ss, sbuf);
    const double nrm = sqrt(ss);
    if (threadIdx.x == 0) norm_out[(long)g * stride_out + off_out] = nrm;
    const double inv = 1.0 / (nrm > 1e-10 ? nrm : 1.0);
    for (int i = threadIdx.x; i < n; i += VT) { const c128 a = s[i]; d[i] = cmake(__dmul_rn(a.x, inv), __dmul_rn(a.y, inv)); }
}

// nrm only
__global__ void __launch_bounds__(VT)
norm_kernel(const c128* __restrict__ S, long ld, const int* __restrict__ slots, int n,
            double* __restrict__ norm_out, int stride_out, int off_out)
{
    __shared__ double sbuf[VT / 64];
    const int g = blockIdx.x;
    const c128* s = S + (long)slots[g] * ld;
    double ss = 0.0;
    for (int i = threadIdx.x; i < n; i += VT) { const c128 a = s[i]; ss = fma(a.x, a.x, ss); ss = fma(a.y, a.y, ss); }
    ss = block_sum(ss, sbuf);
    if (threadIdx.x == 0) norm_out[(long)g * stride_out + off_out] = sqrt(ss);
}

// Hermitian match: scores S[slot][j] = |v^H V[:,j]| (already as complex dots in S) -> argmax (first max),
// then X[slot] <- V[:, idx] / ||V[:, idx]||.
__global__ void __launch_bounds__(VT)
herm_pick_kernel(const c128* __restrict__ S, long lds_, c128* __restrict__ X, long ldx, const int* __restrict__ slots,
                 const c128* __restrict__ V, int n, int* __restrict__ idx_out, double* __restrict__ norm_out)
{
    __shared__ double sval[VT / 64];
    __shared__ int sidx[VT / 64];
    __shared__ double sbuf[VT / 64];
    const int g = blockIdx.x;
    const c128* s = S + (long)slots[g] * lds_;
    double best = -1.0; int bidx = 0x7fffffff;
    for (int j = threadIdx.x; j < n; j += VT) {
        const double v = hypot(s[j].x, s[j].y);
        if (v > best) { best = v; bidx = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double ov = __shfl_xor(best, o, 64); int oi = __shfl_xor(bidx, o, 64);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sval[wave] = best; sidx[wave] = bidx; }
    __syncthreads();
    best = sval[0]; bidx = sidx[0];
#pragma unroll
    for (int w = 1; w < VT / 64; ++w) if (sval[w] > best || (sval[w] == best && sidx[w] < bidx)) { best = sval[w]; bidx = sidx[w]; }
    if (bidx == 0x7fffffff) bidx = 0;       // all-NaN scores: np.argmax returns the first NaN; flagged by the residual
    c128* x = X + (long)slots[g] * ldx;
    double ss = 0.0;
    for (int i = threadIdx.x; i < n; i += VT) { const c128 a = V[(long)i * n + bidx]; ss = fma(a.x, a.x, ss); ss = fma(a.y, a.y, ss); }
    ss = block_sum(ss, sbuf);
    const double nrm = sqrt(ss), inv = 1.0 / nrm;
    for (int i = threadIdx.x; i < n; i += VT) { const c128 a = V[(long)i * n + bidx]; x[i] = cmake(__dmul_rn(a.x, inv), __dmul_rn(a.y, inv)); }
    if (threadIdx.x == 0) { idx_out[g] = bidx; norm_out[g] = nrm; }
}

}  // namespace

void maus_launch_rayleigh_dots(hipStream_t st, const c128* X, const c128* Y, long ld, const int* slots, int count, int n, c128* num, c128* den) {
    hipLaunchKernelGGL(rayleigh_dots_kernel, dim3(count), dim3(VT), 0, st, X, Y, ld, slots, n, num, den);
}
void maus_launch_relax(hipStream_t st, c128* X, const c128* W, long ld, const int* slots, int count, int n, const c128* alpha, int normalise, double* norm_out) {
    hipLaunchKernelGGL(relax_normalise_kernel, dim3(count), dim3(VT), 0, st, X, W, ld, slots, n, alpha, normalise, norm_out);
}
void maus_launch_residual(hipStream_t st, int kind, const c128* X, const c128* Y, long ld, const int* slots, int count, int n,
                          const c128* lam, const c128* bvec, double* resid, int* finite) {
    hipLaunchKernelGGL(residual_kernel, dim3(count), dim3(VT), 0, st, kind, X, Y, ld, slots, n, lam, bvec, resid, finite);
}
void maus_launch_svd_resid(hipStream_t st, const c128* Yv, const c128* Uv, long ld, const int* slots, int count, int n,
                           const c128* sigma, double* out, int accumulate, int* finite) {
    hipLaunchKernelGGL(svd_resid_kernel, dim3(count), dim3(VT), 0, st, Yv, Uv, ld, slots, n, sigma, out, accumulate, finite);
}
void maus_launch_norm_scale(hipStream_t st, const c128* S, c128* D, long ld, const int* slots, int count, int n, double* norm_out, int stride_out, int off_out) {
    hipLaunchKernelGGL(norm_scale_kernel, dim3(count), dim3(VT), 0, st, S, D, ld, slots, n, norm_out, stride_out, off_out);
}
void maus_launch_norm(hipStream_t st, const c128* S, long ld, const int* slots, int count, int n, double* norm_out, int stride_out, int off_out) {
    hipLaunchKernelGGL(norm_kernel, dim3(count), dim3(VT), 0, st, S, ld, slots, n, norm_out, stride_out, off_out);
}
void maus_launch_herm_pick(hipStream_t st, const c128* S, long lds_, c128* X, long ldx, const int* slots, int count, const c128* V, int n, int* idx_out, double* norm_out) {
    hipLaunchKernelGGL(herm_pick_kernel, dim3(count), dim3(VT), 0, st, S, lds_, X, ldx, slots, V, n, idx_out, norm_out);
}
