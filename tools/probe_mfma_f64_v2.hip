// Sustained v_mfma_f64_16x16x4_f64 issue rate with VGPR accumulators (512/1024-thread blocks keep
// the register budget <= 256/128 so the compiler does not place the accumulators in AGPRs).
// Build: hipcc --offload-arch=gfx950 -O3 -o probe_mfma_f64_v2 probe_mfma_f64_v2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NACC, int THREADS>
__global__ void __launch_bounds__(THREADS) k_v(double* out, unsigned long long* clk, int iters, double seed) {
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
    out[(size_t)blockIdx.x * THREADS + l] = s;
    if ((l & 63) == 0) { size_t w = (size_t)blockIdx.x * (THREADS / 64) + (l >> 6); clk[2 * w] = t1 - t0; clk[2 * w + 1] = r1 - r0; }
}

// alternating operands from an array that the compiler cannot fold: random-looking data
template <int NACC, int THREADS>
__global__ void __launch_bounds__(THREADS) k_vdata(const double* __restrict__ in, double* out, int iters) {
    int l = threadIdx.x;
    double a0 = in[l], b0 = in[l + THREADS], a1 = in[l + 2 * THREADS], b1 = in[l + 3 * THREADS];
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64((i & 1) ? a1 : a0, (i & 2) ? b1 : b0, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * THREADS + l] = s;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    double* dOut; CK(hipMalloc(&dOut, sizeof(double) * 1024 * cus * 2));
    unsigned long long* dClk; CK(hipMalloc(&dClk, sizeof(unsigned long long) * 2 * 16 * cus * 2));
    std::vector<double> hin(4096); for (int i = 0; i < 4096; ++i) hin[i] = (double)rand() / RAND_MAX * 2.0 - 1.0;
    double* dIn; CK(hipMalloc(&dIn, 4096 * 8)); CK(hipMemcpy(dIn, hin.data(), 4096 * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](auto launch, const char* name, double flops, int waves_total, bool clk) {
        launch(); CK(hipDeviceSynchronize());
        float best = 1e30f, sum = 0;
        const int reps = 8;
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&sum, e0, e1));
        best = sum / reps;
        printf("%-52s %8.3f ms  %7.2f TFLOP/s", name, best, flops / best * 1e-9);
        if (clk) {
            std::vector<unsigned long long> hc(2 * waves_total);
            CK(hipMemcpy(hc.data(), dClk, sizeof(unsigned long long) * 2 * waves_total, hipMemcpyDeviceToHost));
            double cyc = 0, rt = 0; for (int i = 0; i < waves_total; ++i) { cyc += hc[2 * i]; rt += hc[2 * i + 1]; }
            printf("   clock %.3f GHz (memtime/memrealtime*0.1)  wave-cycles/MFMA %.1f", cyc / rt * 0.1, cyc / waves_total);
        }
        printf("\n");
    };
    const int iters = 100000;
    const double fl = 2.0 * 16 * 16 * 4;
#define RUN(NACC, T, BPC) { char nm[128]; snprintf(nm, sizeof nm, "VGPR acc NACC=%d  %d thr x %d blk/CU = %d waves/SIMD", NACC, T, BPC, T / 256 * BPC); \
        int blocks = cus * BPC; int waves = blocks * (T / 64); \
        run([&] { hipLaunchKernelGGL((k_v<NACC, T>), dim3(blocks), dim3(T), 0, 0, dOut, dClk, iters, 1.0); }, nm, fl * NACC * iters * (double)waves, waves, true); \
        std::vector<unsigned long long> hc(2); }
    RUN(4, 512, 1)      // 2 waves/SIMD
    RUN(8, 512, 1)
    RUN(4, 1024, 1)     // 4 waves/SIMD
    RUN(8, 1024, 1)
    RUN(4, 256, 1)      // 1 wave/SIMD (may use AGPRs)
#define RUND(NACC, T) { char nm[128]; snprintf(nm, sizeof nm, "random operands NACC=%d %d thr (%d waves/SIMD)", NACC, T, T / 256); \
        int blocks = cus; int waves = blocks * (T / 64); \
        run([&] { hipLaunchKernelGGL((k_vdata<NACC, T>), dim3(blocks), dim3(T), 0, 0, dIn, dOut, iters); }, nm, fl * NACC * iters * (double)waves, waves, false); }
    RUND(8, 512)
    RUND(8, 1024)
    return 0;
}
