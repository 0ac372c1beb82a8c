// Which property of the zgemm MFMA group costs issue rate?  Register-only loops replicating the
// 16-MFMA group of zgemm_kernel (8 accumulators, 2 A + 2 B complex fragments).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define MF(a, b, c, neg) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, neg)

// MODE 0: gemm group exactly (neg on Aim*Bim) ; 1: same without neg ; 2: same operand (a0x,b0x) everywhere, 8 acc
// MODE 3: gemm group with ds_read_b128 of next fragments interleaved (LDS traffic, values unused for math)
template <int MODE>
__global__ void __launch_bounds__(512) k_group(const double* __restrict__ in, double* out, int iters) {
    __shared__ double2 lds[2048];
    int l = threadIdx.x;
    for (int i = l; i < 2048; i += 512) lds[i] = make_double2(in[i & 1023], in[(i + 7) & 1023]);
    __syncthreads();
    double a0x = in[l], a0y = in[l + 512], a1x = in[l + 1024], a1y = in[l + 1536];
    double b0x = in[l + 2048], b0y = in[l + 2560], b1x = in[l + 3072], b1y = in[l + 3584];
    d4 c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) {
            double2 t0 = lds[(l + it) & 2047], t1 = lds[(l + it + 64) & 2047], t2 = lds[(l + it + 128) & 2047], t3 = lds[(l + it + 192) & 2047];
            a0x += t0.x * 1e-300; a1x += t1.x * 1e-300; b0x += t2.x * 1e-300; b1x += t3.x * 1e-300;
        }
        if (MODE == 2) {
            MF(a0x, b0x, c0, 0); MF(a0x, b0x, c1, 0); MF(a0x, b0x, c2, 0); MF(a0x, b0x, c3, 0);
            MF(a0x, b0x, c4, 0); MF(a0x, b0x, c5, 0); MF(a0x, b0x, c6, 0); MF(a0x, b0x, c7, 0);
            MF(a0x, b0x, c0, 0); MF(a0x, b0x, c1, 0); MF(a0x, b0x, c2, 0); MF(a0x, b0x, c3, 0);
            MF(a0x, b0x, c4, 0); MF(a0x, b0x, c5, 0); MF(a0x, b0x, c6, 0); MF(a0x, b0x, c7, 0);
        } else if (MODE == 1) {
            MF(a0x, b0x, c0, 0); MF(a0x, b0y, c1, 0); MF(a0x, b1x, c2, 0); MF(a0x, b1y, c3, 0);
            MF(a1x, b0x, c4, 0); MF(a1x, b0y, c5, 0); MF(a1x, b1x, c6, 0); MF(a1x, b1y, c7, 0);
            MF(a0y, b0y, c0, 0); MF(a0y, b0x, c1, 0); MF(a0y, b1y, c2, 0); MF(a0y, b1x, c3, 0);
            MF(a1y, b0y, c4, 0); MF(a1y, b0x, c5, 0); MF(a1y, b1y, c6, 0); MF(a1y, b1x, c7, 0);
        } else {
            MF(a0x, b0x, c0, 0); MF(a0x, b0y, c1, 0); MF(a0x, b1x, c2, 0); MF(a0x, b1y, c3, 0);
            MF(a1x, b0x, c4, 0); MF(a1x, b0y, c5, 0); MF(a1x, b1x, c6, 0); MF(a1x, b1y, c7, 0);
            MF(a0y, b0y, c0, 1); MF(a0y, b0x, c1, 0); MF(a0y, b1y, c2, 1); MF(a0y, b1x, c3, 0);
            MF(a1y, b0y, c4, 1); MF(a1y, b0x, c5, 0); MF(a1y, b1y, c6, 1); MF(a1y, b1x, c7, 0);
        }
    }
    d4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    out[(size_t)blockIdx.x * 512 + l] = s[0] + s[1] + s[2] + s[3];
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    std::vector<double> hin(4096); for (int i = 0; i < 4096; ++i) hin[i] = (double)rand() / RAND_MAX - 0.5;
    double *dIn, *dOut; CK(hipMalloc(&dIn, 4096 * 8)); CK(hipMalloc(&dOut, 8 * 512 * cus));
    CK(hipMemcpy(dIn, hin.data(), 4096 * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    auto run = [&](auto launch, const char* name) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int r = 0; r < 4; ++r) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 4;
        double fl = 2.0 * 16 * 16 * 4 * 16.0 * iters * 8.0 * cus;
        printf("%-56s %8.3f ms  %7.2f TFLOP/s\n", name, ms, fl / ms * 1e-9);
    };
    run([&] { hipLaunchKernelGGL(k_group<2>, dim3(cus), dim3(512), 0, 0, dIn, dOut, iters); }, "8 acc, one operand pair");
    run([&] { hipLaunchKernelGGL(k_group<1>, dim3(cus), dim3(512), 0, 0, dIn, dOut, iters); }, "gemm operand pattern, no neg");
    run([&] { hipLaunchKernelGGL(k_group<0>, dim3(cus), dim3(512), 0, 0, dIn, dOut, iters); }, "gemm operand pattern, neg on 4 of 16 (as zgemm)");
    run([&] { hipLaunchKernelGGL(k_group<3>, dim3(cus), dim3(512), 0, 0, dIn, dOut, iters); }, "as zgemm + 4 ds_read_b128 per group");
    return 0;
}
