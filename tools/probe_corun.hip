// Can bandwidth-bound kernels hide behind the MFMA-bound trailing update when both run on separate streams?
// The zgemm of the LU trailing update (4 waves x 128 VGPRs per workgroup, 4 workgroups per CU) is launched on one
// stream and a streaming read-modify-write kernel on another; prints each alone, both together, and how much of the
// shorter one was hidden.  Co-runner variants differ in workgroup footprint (what the dispatcher must find free):
//   thin256 : 256 threads, <= 64 VGPRs     (laswp / trsm / build_h class)
//   mid256  : 256 threads, <= 128 VGPRs    (a zgemm-sized footprint)
//   half512 : 512 threads, <= 128 VGPRs    (half a CU)
//   fat512  : 512 threads, ~240 VGPRs      (whole CU: the present lu_panel)
// Build: hipcc --offload-arch=gfx950 -O3 -I adaptive_matrix_solver_amd/csrc -o tools/bin/probe_corun tools/probe_corun.hip
#include "../adaptive_matrix_solver_amd/csrc/zgemm.hip"
#include <cstdio>
#include <vector>
#include <chrono>

template <int NREG>
__device__ __forceinline__ void rmw(double2* __restrict__ p, size_t n, size_t stride_elems) {
    // each thread streams NREG/4 independent 16-B elements per trip (loads first, then stores)
    constexpr int U = NREG / 4;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i + (U - 1) * step < n; i += U * step) {
        double2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * step];
#pragma unroll
        for (int u = 0; u < U; ++u) { v[u].x = v[u].x * 1.0000001 + 1e-9; v[u].y -= 1e-9; }
#pragma unroll
        for (int u = 0; u < U; ++u) p[i + u * step] = v[u];
    }
}
__global__ void __launch_bounds__(256, 8) thin256(double2* p, size_t n) { rmw<16>(p, n, 0); }
__global__ void __launch_bounds__(256, 4) mid256(double2* p, size_t n) { rmw<96>(p, n, 0); }
__global__ void __launch_bounds__(512, 4) half512(double2* p, size_t n) { rmw<96>(p, n, 0); }
__global__ void __launch_bounds__(512, 2) fat512(double2* p, size_t n) { rmw<192>(p, n, 0); }

__global__ void fill(double* p, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) { unsigned x = (unsigned)i * 2654435761u + 12345u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; p[i] = (double)(x & 0xffffff) / 16777216.0 - 0.5; }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 136;           // matrices in the GEMM batch
    const int M = 3584, N = 3584 + 32, K = 512, ld = 4128;
    const long rows = (long)M + K;
    const size_t per = (size_t)rows * ld;
    c128* base; CK(hipMalloc((void**)&base, sizeof(c128) * per * G));
    const size_t nmem = (size_t)1 << 30;                    // 16 GiB of c128 for the co-runner (read + write = 32 GiB of traffic)
    double2* mem; CK(hipMalloc((void**)&mem, sizeof(double2) * nmem));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (double*)base, per * G * 2);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (double*)mem, nmem * 2);
    CK(hipDeviceSynchronize());
    int plo = 0, phi = 0; (void)hipDeviceGetStreamPriorityRange(&plo, &phi);
    hipStream_t sg, sm; CK(hipStreamCreateWithPriority(&sg, hipStreamNonBlocking, plo)); CK(hipStreamCreateWithPriority(&sm, hipStreamNonBlocking, phi));
    hipEvent_t g0, g1, m0, m1, o; hipEventCreate(&g0); hipEventCreate(&g1); hipEventCreate(&m0); hipEventCreate(&m1); hipEventCreate(&o);
    auto gemm = [&](hipStream_t st) {
        maus_zgemm_launch(st, M, N, K, base + (size_t)K * ld, ld, (long)per, base + K, ld, (long)per, base + (size_t)K * ld + K, ld, (long)per,
                          -1.0, 1, G, 0, false, false);
    };
    struct Var { const char* name; int threads; void (*k)(double2*, size_t); };
    Var vars[] = {{"thin256", 256, thin256}, {"mid256", 256, mid256}, {"half512", 512, half512}, {"fat512", 512, fat512}};
    // warm up
    gemm(sg); for (auto& v : vars) hipLaunchKernelGGL(v.k, dim3(2048), dim3(v.threads), 0, sm, mem, nmem >> 4);
    CK(hipDeviceSynchronize());
    float tg = 0;
    { hipEventRecord(g0, sg); gemm(sg); gemm(sg); hipEventRecord(g1, sg); hipEventSynchronize(g1); hipEventElapsedTime(&tg, g0, g1); }
    printf("zgemm %dx%dx%d batch %d, two launches alone: %.2f ms (%.1f TFLOP/s algorithmic)\n", M, N, K, G, tg, 2 * 8.0 * M * N * K * G / (tg * 1e-3) / 1e12);
    // The co-runner is launched ~25 ms AFTER the first of two back-to-back GEMM launches (the chip is saturated with zgemm
    // workgroups by then), as `pieces` consecutive launches that together stream 4 x 32 GiB.
    for (int grid : {256, 512, 1024}) for (int pieces : {1, 16}) for (auto& v : vars) {
        if (v.threads == 512) continue;          // half- and whole-CU workgroups only start once the zgemm grid has drained (first version of this probe)
        const size_t n = nmem;
        const int reps = 3;
        float tm = 0, tb_g = 0, tb_m = 0, e_g = 0, e_m = 0, t_m_start = 0;
        auto mem_work = [&]() {
            for (int r = 0; r < reps; ++r)
                for (int pc = 0; pc < pieces; ++pc)
                    hipLaunchKernelGGL(v.k, dim3(grid), dim3(v.threads), 0, sm, mem + (n / pieces) * pc, n / pieces);
        };
        hipEventRecord(m0, sm); mem_work(); hipEventRecord(m1, sm);
        hipEventSynchronize(m1); hipEventElapsedTime(&tm, m0, m1);
        CK(hipDeviceSynchronize());
        hipEventRecord(o, sg);
        hipEventRecord(g0, sg); gemm(sg); gemm(sg); hipEventRecord(g1, sg);
        { auto t0 = std::chrono::steady_clock::now(); while (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() < 25.0) {} }
        hipEventRecord(m0, sm); mem_work(); hipEventRecord(m1, sm);
        CK(hipDeviceSynchronize());
        hipEventElapsedTime(&tb_g, g0, g1); hipEventElapsedTime(&tb_m, m0, m1);
        hipEventElapsedTime(&e_g, o, g1); hipEventElapsedTime(&e_m, o, m1); hipEventElapsedTime(&t_m_start, o, m0);
        const float wall = e_g > e_m ? e_g : e_m;
        const double bytes = 32.0 * n * reps;
        printf("%-8s grid %4d x%2d launches/rep %6.1f GB: alone %.2f ms (%.2f TB/s) | launched at %.1f ms into the gemm pair: gemm %.2f ms (alone %.2f), mem took %.2f ms (ends at %.1f), wall %.2f | hidden %.0f %% of the co-runner\n",
               v.name, grid, pieces, bytes / 1e9, tm, bytes / (tm * 1e-3) / 1e12, t_m_start, tb_g, tg, tb_m, e_m, wall,
               100.0 * (tg + tm - wall) / tm);
    }
    return 0;
}
