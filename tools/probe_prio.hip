// Can a high-priority stream's "fat" kernel (8 waves x ~240 VGPRs per workgroup = a whole CU, like lu_panel) get
// its workgroups placed while a low-priority stream keeps the chip saturated with small, short workgroups (4 waves
// x 128 VGPRs, like the zgemm)?  Prints when the fat kernel finished relative to the thin grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>

template <int NREG>
__device__ __forceinline__ double burn(double seed, int iters) {
    double r[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) r[i] = seed + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NREG; ++i) r[i] = fma(r[i], 1.0000001, r[(i + 1) % NREG] * 1e-9);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NREG; ++i) s += r[i];
    return s;
}

__global__ void __launch_bounds__(256, 4) thin_kernel(double* out, int iters) {       // <= 128 VGPRs, 4 WG/CU
    double s = burn<48>(threadIdx.x * 1e-3, iters);
    if (s == 12345.678) out[blockIdx.x] = s;
}
__global__ void __launch_bounds__(512, 2) fat_kernel(double* out, int iters) {        // ~240 VGPRs, 1 WG/CU
    double s = burn<112>(threadIdx.x * 1e-3, iters);
    if (s == 12345.678) out[blockIdx.x] = s;
}

__global__ void __launch_bounds__(512, 4) half_kernel(double* out, int iters) {       // <= 128 VGPRs, 8 waves: half a CU
    double s = burn<48>(threadIdx.x * 1e-3, iters);
    if (s == 12345.678) out[blockIdx.x] = s;
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    printf("priority range: least %d, greatest %d\n", lo, hi);
    double* d; hipMalloc(&d, 1 << 24);
    for (int mode = 0; mode < 3; ++mode) {
        hipStream_t sThin, sFat;
        hipStreamCreateWithPriority(&sThin, hipStreamNonBlocking, mode == 2 ? lo : 0);
        hipStreamCreateWithPriority(&sFat, hipStreamNonBlocking, mode >= 1 ? hi : 0);
        hipEvent_t eThin, eFat; hipEventCreate(&eThin); hipEventCreate(&eFat);
        // warm up
        hipLaunchKernelGGL(thin_kernel, dim3(1024), dim3(256), 0, sThin, d, 10);
        hipLaunchKernelGGL(fat_kernel, dim3(136), dim3(512), 0, sFat, d, 10);
        hipDeviceSynchronize();
        // fat alone
        double t0 = now_ms();
        hipLaunchKernelGGL(fat_kernel, dim3(136), dim3(512), 0, sFat, d, 2000);
        hipStreamSynchronize(sFat);
        double fat_alone = now_ms() - t0;
        // thin alone
        t0 = now_ms();
        hipLaunchKernelGGL(thin_kernel, dim3(200000), dim3(256), 0, sThin, d, 600);
        hipStreamSynchronize(sThin);
        double thin_alone = now_ms() - t0;
        // together: thin first, fat 2 ms later
        t0 = now_ms();
        hipLaunchKernelGGL(thin_kernel, dim3(200000), dim3(256), 0, sThin, d, 600);
        hipEventRecord(eThin, sThin);
        while (now_ms() - t0 < 2.0) { }
        double tf0 = now_ms();
        hipLaunchKernelGGL(fat_kernel, dim3(136), dim3(512), 0, sFat, d, 2000);
        hipEventRecord(eFat, sFat);
        hipEventSynchronize(eFat);
        double fat_done = now_ms() - tf0;
        hipEventSynchronize(eThin);
        double all_done = now_ms() - t0;
        printf("mode %d (fat prio %s, thin prio %s): fat alone %.2f ms, thin alone %.2f ms; together: fat finished %.2f ms after its launch, both done after %.2f ms\n",
               mode, mode >= 1 ? "high" : "normal", mode == 2 ? "low" : "normal", fat_alone, thin_alone, fat_done, all_done);
        // the same with half-CU workgroups (8 waves x <= 128 VGPRs)
        t0 = now_ms();
        hipLaunchKernelGGL(half_kernel, dim3(136), dim3(512), 0, sFat, d, 4000);
        hipStreamSynchronize(sFat);
        double half_alone = now_ms() - t0;
        t0 = now_ms();
        hipLaunchKernelGGL(thin_kernel, dim3(200000), dim3(256), 0, sThin, d, 600);
        hipEventRecord(eThin, sThin);
        while (now_ms() - t0 < 2.0) { }
        tf0 = now_ms();
        hipLaunchKernelGGL(half_kernel, dim3(136), dim3(512), 0, sFat, d, 4000);
        hipEventRecord(eFat, sFat);
        hipEventSynchronize(eFat);
        double half_done = now_ms() - tf0;
        hipEventSynchronize(eThin);
        printf("        half-CU workgroups: alone %.2f ms; beside the thin grid finished %.2f ms after launch, both done after %.2f ms\n",
               half_alone, half_done, now_ms() - t0);
        hipStreamDestroy(sThin); hipStreamDestroy(sFat);
    }
    return 0;
}
