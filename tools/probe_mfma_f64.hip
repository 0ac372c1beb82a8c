// Hardware probe for v_mfma_f64_16x16x4_f64 on gfx950: operand/result lane maps,
// BLGP-as-negate behaviour, issue rate, plus fp64 VALU FMA rate and a stream copy.
// Build: hipcc --offload-arch=gfx950 -O3 -o probe_mfma_f64 probe_mfma_f64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef double d4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int BLGP>
__global__ void k_layout(const double* A, const double* B, double* D) {
    // A: 16x4 row-major, B: 4x16 row-major.  Assumed operand map: a = A[l&15][l>>4], b = B[l>>4][l&15]
    int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];
    double b = B[(l >> 4) * 16 + (l & 15)];
    d4 c = {1000.0, 1000.0, 1000.0, 1000.0};
    d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, BLGP);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = d[r];
}

template <int NACC>
__global__ void __launch_bounds__(256) k_rate(double* out, int iters, double seed) {
    int l = threadIdx.x;
    double a = seed + l * 1e-3, b = seed - l * 1e-3;
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + l] = s;
}

template <int NACC>
__global__ void __launch_bounds__(256) k_rate_clk(double* out, unsigned long long* clk, int iters, double seed) {
    int l = threadIdx.x;
    double a = seed + l * 1e-3, b = seed - l * 1e-3;
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + l] = s;
    if (l == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

// co-execution: in each block of 8 waves (2 per SIMD) waves 0-3 issue MFMA f64, waves 4-7 VALU f64 FMA
__global__ void __launch_bounds__(512) k_coexec(double* out, int iters_mfma, int iters_valu, double seed) {
    int l = threadIdx.x;
    if ((l >> 8) == 0) {
        double a = seed + l * 1e-3, b = seed - l * 1e-3;
        d4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
        for (int it = 0; it < iters_mfma; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        double s = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        out[blockIdx.x * blockDim.x + l] = s;
    } else {
        double x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = seed + i + l * 1e-3;
        double m = 1.0000001, c = 1e-9;
        for (int it = 0; it < iters_valu; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = fma(x[i], m, c);
        }
        double s = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += x[i];
        out[blockIdx.x * blockDim.x + l] = s;
    }
}

__global__ void __launch_bounds__(256) k_valu(double* out, int iters, double seed) {
    int l = threadIdx.x;
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = seed + i + l * 1e-3;
    double m = 1.0000001, c = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = fma(x[i], m, c);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + l] = s;
}

__global__ void __launch_bounds__(256) k_copy(const double2* __restrict__ in, double2* __restrict__ out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

template <int BLGP>
static void run_layout(const std::vector<double>& hA, const std::vector<double>& hB, double* dA, double* dB, double* dD) {
    std::vector<double> hD(256);
    hipLaunchKernelGGL(k_layout<BLGP>, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CK(hipMemcpy(hD.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
    // reference products
    double P[16][16];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j]; P[i][j] = s; }
    int okF64 = 0, okF32 = 0, okNegAB = 0, okNegC = 0, okNegBoth = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        double got = hD[l * 4 + r];
        int col = l & 15, row64 = (l >> 4) + 4 * r, row32 = (l >> 4) * 4 + r;
        if (got == 1000.0 + P[row64][col]) okF64++;
        if (got == 1000.0 + P[row32][col]) okF32++;
        if (got == 1000.0 - P[row64][col]) okNegAB++;
        if (got == -1000.0 + P[row64][col]) okNegC++;
        if (got == -1000.0 - P[row64][col]) okNegBoth++;
    }
    printf("BLGP=%d: match f64-map(+AB+C)=%d/256  f32-map=%d/256  (-AB+C)=%d  (+AB-C)=%d  (-AB-C)=%d\n", BLGP, okF64, okF32, okNegAB, okNegC, okNegBoth);
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d kHz  mem=%.1f GB\n", prop.name, prop.multiProcessorCount, prop.clockRate, prop.totalGlobalMem / 1e9);
    std::vector<double> hA(64), hB(64);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) hA[i * 4 + k] = 1 + i + 17 * k;          // asymmetric integers
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = 3 + 5 * j + 101 * k + (j == 3 ? 7 : 0);
    double *dA, *dB, *dD;
    CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dD, 256 * 8));
    CK(hipMemcpy(dA, hA.data(), 64 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), 64 * 8, hipMemcpyHostToDevice));
    run_layout<0>(hA, hB, dA, dB, dD);
    run_layout<1>(hA, hB, dA, dB, dD);
    run_layout<2>(hA, hB, dA, dB, dD);
    run_layout<3>(hA, hB, dA, dB, dD);
    run_layout<4>(hA, hB, dA, dB, dD);

    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double* dOut; CK(hipMalloc(&dOut, sizeof(double) * 256 * 2048 * 4));
    int cus = prop.multiProcessorCount;
    auto time_it = [&](auto launch, const char* name, double flops) {
        launch(); CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
        printf("%-44s %8.3f ms  %8.2f TFLOP/s\n", name, best, flops / best * 1e-9);
    };
    const int iters = 20000;
    for (int wps = 1; wps <= 2; ++wps) {   // waves per SIMD: block = 256 threads = 4 waves = 1 wave/SIMD
        int blocks = cus * wps;
        double fl = 2.0 * 16 * 16 * 4;  // per MFMA
        char nm[128];
        snprintf(nm, sizeof nm, "mfma_f64_16x16x4 NACC=1 %d wave/SIMD", wps);
        time_it([&] { hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, 0, dOut, iters, 1.0); }, nm, fl * 1 * iters * 4.0 * blocks);
        snprintf(nm, sizeof nm, "mfma_f64_16x16x4 NACC=2 %d wave/SIMD", wps);
        time_it([&] { hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(256), 0, 0, dOut, iters, 1.0); }, nm, fl * 2 * iters * 4.0 * blocks);
        snprintf(nm, sizeof nm, "mfma_f64_16x16x4 NACC=4 %d wave/SIMD", wps);
        time_it([&] { hipLaunchKernelGGL(k_rate<4>, dim3(blocks), dim3(256), 0, 0, dOut, iters, 1.0); }, nm, fl * 4 * iters * 4.0 * blocks);
        snprintf(nm, sizeof nm, "mfma_f64_16x16x4 NACC=8 %d wave/SIMD", wps);
        time_it([&] { hipLaunchKernelGGL(k_rate<8>, dim3(blocks), dim3(256), 0, 0, dOut, iters, 1.0); }, nm, fl * 8 * iters * 4.0 * blocks);
    }
    {
        unsigned long long* dClk; CK(hipMalloc(&dClk, sizeof(unsigned long long) * 2 * cus * 8));
        for (int wps = 1; wps <= 8; wps *= 2) {
            int blocks = cus * wps;
            char nm[128];
            snprintf(nm, sizeof nm, "mfma_f64 NACC=4 (alt neg) %d wave/SIMD +clk", wps);
            time_it([&] { hipLaunchKernelGGL(k_rate_clk<4>, dim3(blocks), dim3(256), 0, 0, dOut, dClk, iters, 1.37); }, nm, 2.0 * 16 * 16 * 4 * 4 * iters * 4.0 * blocks);
            std::vector<unsigned long long> hc(2 * blocks);
            CK(hipMemcpy(hc.data(), dClk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost));
            double cyc = 0, rt = 0; for (int i = 0; i < blocks; ++i) { cyc += hc[2 * i]; rt += hc[2 * i + 1]; }
            printf("    in-kernel clock = %.3f GHz; shader cycles per MFMA per wave = %.1f\n", cyc / rt * 0.1, cyc / blocks / (4.0 * iters));
        }
    }
    {
        // 1 MFMA wave + 1 VALU wave per SIMD; iterations balanced so both halves run about equally long
        for (int ratio : {0, 8, 16, 24}) {
            int im = 20000, iv = im * ratio;
            char nm[128];
            snprintf(nm, sizeof nm, "coexec: MFMA x4 + VALU (%d fma-iters per mfma-iter)", ratio);
            double fl = (2.0 * 16 * 16 * 4 * 4 * im * 4.0 + 2.0 * 8 * (double)iv * 256.0) * cus;
            time_it([&] { hipLaunchKernelGGL(k_coexec, dim3(cus), dim3(512), 0, 0, dOut, im, iv, 1.0); }, nm, fl);
        }
    }
    for (int wps = 1; wps <= 4; wps *= 2) {
        int blocks = cus * wps;
        char nm[128];
        snprintf(nm, sizeof nm, "v_fma_f64 x8 indep  %d wave/SIMD", wps);
        time_it([&] { hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, 0, dOut, iters, 1.0); }, nm, 2.0 * 8 * iters * 256.0 * blocks);
    }
    // cycles per MFMA estimate at nominal clock
    {
        size_t n = (size_t)1 << 28;  // 4 GiB of double2 in + 4 GiB out
        double2 *din, *dout; CK(hipMalloc(&din, n * 16)); CK(hipMalloc(&dout, n * 16));
        CK(hipMemset(din, 1, n * 16));
        auto launch = [&] { hipLaunchKernelGGL(k_copy, dim3(cus * 8), dim3(256), 0, 0, din, dout, n); };
        launch(); CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
        printf("stream copy 4GiB->4GiB: %.3f ms  %.2f TB/s (read+write)\n", best, 2.0 * n * 16 / best * 1e-9);
        CK(hipFree(din)); CK(hipFree(dout));
    }
    return 0;
}
