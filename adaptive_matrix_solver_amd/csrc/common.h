// Shared device/host helpers for libmaus_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double2 c128;                                   // interleaved (re, im) = NumPy complex128
typedef double d4 __attribute__((ext_vector_type(4)));  // one v_mfma_f64_16x16x4 accumulator

#define MAUS_WAVE 64

__device__ __forceinline__ c128 cmake(double r, double i) { c128 z; z.x = r; z.y = i; return z; }
__device__ __forceinline__ c128 cadd(c128 a, c128 b) { return cmake(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ c128 csub(c128 a, c128 b) { return cmake(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ c128 cmul(c128 a, c128 b) { return cmake(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ c128 cconj(c128 a) { return cmake(a.x, -a.y); }
// a -= b*c
__device__ __forceinline__ void cfms(c128& a, c128 b, c128 c) {
    a.x = fma(-b.x, c.x, a.x); a.x = fma(b.y, c.y, a.x);
    a.y = fma(-b.x, c.y, a.y); a.y = fma(-b.y, c.x, a.y);
}
// a += b*c
__device__ __forceinline__ void cfma(c128& a, c128 b, c128 c) {
    a.x = fma(b.x, c.x, a.x); a.x = fma(-b.y, c.y, a.x);
    a.y = fma(b.x, c.y, a.y); a.y = fma(b.y, c.x, a.y);
}
// a += conj(b)*c
__device__ __forceinline__ void cfma_conj(c128& a, c128 b, c128 c) {
    a.x = fma(b.x, c.x, a.x); a.x = fma(b.y, c.y, a.x);
    a.y = fma(b.x, c.y, a.y); a.y = fma(-b.y, c.x, a.y);
}
// LAPACK dcabs1: |re| + |im|  (izamax pivot rule)
__device__ __forceinline__ double cabs1(c128 a) { return fabs(a.x) + fabs(a.y); }
// 1/z by Smith's algorithm (no intermediate overflow for representable results)
__device__ __forceinline__ c128 crecip(c128 z) {
    if (fabs(z.x) >= fabs(z.y)) {
        double t = z.y / z.x, d = z.x + z.y * t;
        return cmake(1.0 / d, -t / d);
    } else {
        double t = z.x / z.y, d = z.x * t + z.y;
        return cmake(t / d, -1.0 / d);
    }
}
// a / b by Smith's algorithm
__device__ __forceinline__ c128 cdiv(c128 a, c128 b) {
    if (fabs(b.x) >= fabs(b.y)) {
        double t = b.y / b.x, d = b.x + b.y * t;
        return cmake((a.x + a.y * t) / d, (a.y - a.x * t) / d);
    } else {
        double t = b.x / b.y, d = b.x * t + b.y;
        return cmake((a.x * t + a.y) / d, (a.y * t - a.x) / d);
    }
}
__device__ __forceinline__ bool cfinite(c128 a) { return isfinite(a.x) && isfinite(a.y); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum over the 64 lanes on the DPP network, the same value in every lane, in a fixed order: four steps inside each row of 16
// lanes (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror: the lanes of a half already agree, so a mirror adds the
// other half), then the four rows.  wave_sum above goes through ds_bpermute six times.
__device__ __forceinline__ double wave_sum_dpp(double v) {
#define MAUS_DPP_F64S(CTRL) __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false), \
                                             __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false))
    v += MAUS_DPP_F64S(0xB1);
    v += MAUS_DPP_F64S(0x4E);
    v += MAUS_DPP_F64S(0x141);
    v += MAUS_DPP_F64S(0x140);
#undef MAUS_DPP_F64S
    auto lane_f64 = [&](int l) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l)); };
    return (lane_f64(0) + lane_f64(16)) + (lane_f64(32) + lane_f64(48));
}

// Pivot search inside one wave (LAPACK izamax: largest value, lowest index among equals): on return every lane holds the wave's
// (best, bidx).  Values are >= 0 or -1 ("no row"), never NaN.  Two reductions on the DPP network -- the maximum of the values,
// then the minimum index among the lanes that hold it -- instead of six butterfly rounds through ds_bpermute (three LDS-crossbar
// trips per round, each waited for): quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror make the 16 lanes of a DPP row
// agree (max and min are idempotent, so a mirror is as good as a butterfly), four v_readlane join the rows.
__device__ __forceinline__ void wave_argmax(double& best, int& bidx) {
    double v = best;
#define MAUS_DPP_F64(CTRL) __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false), \
                                            __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false))
    v = fmax(v, MAUS_DPP_F64(0xB1));
    v = fmax(v, MAUS_DPP_F64(0x4E));
    v = fmax(v, MAUS_DPP_F64(0x141));
    v = fmax(v, MAUS_DPP_F64(0x140));
#undef MAUS_DPP_F64
    auto lane_f64 = [&](int l) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l)); };
    const double m = fmax(fmax(lane_f64(0), lane_f64(16)), fmax(lane_f64(32), lane_f64(48)));
    unsigned c = (best == m) ? (unsigned)bidx : 0xffffffffu;             // bidx >= 0 (INT_MAX = none)
    c = min(c, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)c, 0xB1, 0xf, 0xf, false));
    c = min(c, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)c, 0x4E, 0xf, 0xf, false));
    c = min(c, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)c, 0x141, 0xf, 0xf, false));
    c = min(c, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)c, 0x140, 0xf, 0xf, false));
    const unsigned i0 = (unsigned)__builtin_amdgcn_readlane((int)c, 0), i1 = (unsigned)__builtin_amdgcn_readlane((int)c, 16);
    const unsigned i2 = (unsigned)__builtin_amdgcn_readlane((int)c, 32), i3 = (unsigned)__builtin_amdgcn_readlane((int)c, 48);
    best = m;
    bidx = (int)min(min(i0, i1), min(i2, i3));
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for vmcnt(0), i.e. for every global
// load and STORE the wave has in flight; where only LDS data changes hands between the phases that is a needless
// HBM round trip per barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// kernel classes for the profiling counters (maus_profile_read)
enum { KC_GEMM = 0, KC_PANEL = 1, KC_TRSM = 2, KC_LASWP = 3, KC_BUILD = 4, KC_BACKSOLVE = 5, KC_VEC = 6,
       KC_GEMM_K128 = 7, KC_GEMM_K64 = 8, KC_GEMM_K32 = 9, KC_GEMM_K16 = 10, KC_COUNT = 11 };   // 7..10: LU recursion GEMMs by K (<256)
