// Device-resident timing of the batched LU-update GEMM shape (measurement tool, not part of libmaus_hip: until round 3 this was
// an entry point of the product's C ABI).  C[M,N] -= A[M,K] B[K,N] on `batch` matrices, each embedded in one row-major array of
// leading dimension ld like the round-1 LU workspace (A = rows K.., cols 0..K; B = rows 0..K, cols K..; C = rows K.., cols K..).
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/bin/zgemm_bench tools/zgemm_bench.hip
//   tools/bin/zgemm_bench M N K ld batch [iters] [zero] [shared]
//       zero    all-zero operands (clock check: the chip holds a higher clock on zeros)
//       shared  every matrix reads the same A and B (traffic check)
//       pop     a population product  C[M,N] = A[M,K] * op(B):  `pop` alone B as [N][K] (dot-product layout, the matvec Y = X A^T of
//               AMS:264 / 295), `pop plain` B as [K][N]; `conja` / `conjb` conjugate an operand.  One matrix (batch = 1), ld ignored.
//       lu      the LU's own operand form: tile-major H / U arrays of npad = M + K rows (luws.h), A and C rows through a row list
//               (identity; `perm`: a random permutation of the rows below K, as implicit pivoting leaves them)
// Prints ms per launch and 8MNK-equivalent TFLOP/s.  The kernels come straight from the library's translation unit.
#include "../adaptive_matrix_solver_amd/csrc/zgemm.hip"
#include <cstdio>
#include <cstring>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
#include "../adaptive_matrix_solver_amd/csrc/luws.h"

__global__ void fill_rand_kernel(double* p, size_t n, unsigned seed) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) { unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; p[i] = (double)(x & 0xffffff) / 16777216.0 - 0.5; }
}

#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s M N K ld batch [iters] [zero] [shared]\n", argv[0]); return 2; }
    const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]), ld = atoi(argv[4]), batch = atoi(argv[5]);
    const int iters = argc > 6 ? atoi(argv[6]) : 5;
    bool zero = false, shared = false, lu = false, permute = false;
    for (int i = 7; i < argc; ++i) { zero |= !strcmp(argv[i], "zero"); shared |= !strcmp(argv[i], "shared"); lu |= !strcmp(argv[i], "lu"); permute |= !strcmp(argv[i], "perm"); }
    bool pop = false, plain = false, conja = false, conjb = false;
    for (int i = 7; i < argc; ++i) { pop |= !strcmp(argv[i], "pop"); plain |= !strcmp(argv[i], "plain"); conja |= !strcmp(argv[i], "conja"); conjb |= !strcmp(argv[i], "conjb"); }
    if (pop) {
        c128 *A = nullptr, *B = nullptr, *C = nullptr;
        CK(hipMalloc((void**)&A, sizeof(c128) * (size_t)M * K));
        CK(hipMalloc((void**)&B, sizeof(c128) * (size_t)N * K));
        CK(hipMalloc((void**)&C, sizeof(c128) * (size_t)M * N));
        hipStream_t st; CK(hipStreamCreate(&st));
        hipLaunchKernelGGL(fill_rand_kernel, dim3(2048), dim3(256), 0, st, (double*)A, (size_t)M * K * 2, 12345u);
        hipLaunchKernelGGL(fill_rand_kernel, dim3(2048), dim3(256), 0, st, (double*)B, (size_t)N * K * 2, 777u);
        const int blay = plain ? 0 : 1;
        auto launch = [&]() { maus_zgemm_launch(st, M, N, K, A, K, 0, B, plain ? N : K, 0, C, N, 0, 1.0, 0, 1, blay, conja, conjb); };
        launch();
        CK(hipStreamSynchronize(st));
        hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
        CK(hipEventRecord(t0, st));
        for (int i = 0; i < iters; ++i) launch();
        CK(hipEventRecord(t1, st));
        CK(hipEventSynchronize(t1));
        float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
        ms /= iters;
        CK(hipGetLastError());
        printf("M=%d N=%d K=%d pop%s%s%s: %.3f ms per launch, %.1f TFLOP/s (8MNK)\n", M, N, K, plain ? " plain" : "", conja ? " conja" : "", conjb ? " conjb" : "",
               ms, 8.0 * M * N * K / (ms * 1e-3) / 1e12);
        return 0;
    }
    if (lu) {
        // H[rows[m]][K + n] -= H[rows[m]][k] * U[k][K + n]: the first trailing update of a matrix of npad = M + K rows, N <= M + 32
        const int npad = M + K;
        if (N > M + 32 || (npad % 64)) { fprintf(stderr, "lu mode: N <= M + 32, M + K a multiple of 64\n"); return 2; }
        const long strideH = (long)npad * lu_ntiles(npad) * LU_TW;
        c128 *H = nullptr, *U = nullptr; int* rows = nullptr;
        CK(hipMalloc((void**)&H, sizeof(c128) * strideH * batch));
        CK(hipMalloc((void**)&U, sizeof(c128) * strideH * batch));
        CK(hipMalloc((void**)&rows, sizeof(int) * (size_t)npad * batch));
        hipStream_t st; CK(hipStreamCreate(&st));
        hipLaunchKernelGGL(fill_rand_kernel, dim3(2048), dim3(256), 0, st, (double*)H, (size_t)strideH * batch * 2, 12345u);
        hipLaunchKernelGGL(fill_rand_kernel, dim3(2048), dim3(256), 0, st, (double*)U, (size_t)strideH * batch * 2, 777u);
        std::vector<int> hr((size_t)npad * batch);
        std::mt19937 gen(1);
        for (int g = 0; g < batch; ++g) {
            int* r = hr.data() + (size_t)g * npad;
            std::iota(r, r + npad, 0);
            if (permute) std::shuffle(r, r + npad, gen);      // after K pivot steps the remaining rows are anywhere
        }
        CK(hipMemcpy(rows, hr.data(), sizeof(int) * hr.size(), hipMemcpyHostToDevice));
        auto launch = [&]() { maus_zgemm_launch_lu(st, M, N, K, H, U, H, npad, strideH, 0, 0, K, batch, rows + K, npad); };
        launch();
        CK(hipStreamSynchronize(st));
        hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
        CK(hipEventRecord(t0, st));
        for (int i = 0; i < iters; ++i) launch();
        CK(hipEventRecord(t1, st));
        CK(hipEventSynchronize(t1));
        float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
        ms /= iters;
        CK(hipGetLastError());
        printf("M=%d N=%d K=%d npad=%d batch=%d lu%s: %.3f ms per launch, %.1f TFLOP/s (8MNK)\n", M, N, K, npad, batch, permute ? " perm" : "",
               ms, 8.0 * M * N * K * batch / (ms * 1e-3) / 1e12);
        return 0;
    }
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0 || iters <= 0 || ld < (N > K ? N : K) + K) { fprintf(stderr, "bad sizes\n"); return 2; }
    const size_t per = (size_t)((long)M + K) * ld;
    c128* base = nullptr;
    CK(hipMalloc((void**)&base, sizeof(c128) * per * batch));
    hipStream_t st; CK(hipStreamCreate(&st));
    if (zero) CK(hipMemsetAsync(base, 0, sizeof(c128) * per * batch, st));
    else hipLaunchKernelGGL(fill_rand_kernel, dim3(2048), dim3(256), 0, st, (double*)base, per * batch * 2, 12345u);
    const long sAB = shared ? 0 : (long)per;
    auto launch = [&]() {
        maus_zgemm_launch(st, M, N, K, base + (size_t)K * ld, ld, sAB, base + K, ld, sAB, base + (size_t)K * ld + K, ld, (long)per,
                          -1.0, 1, batch, 0, false, false);
    };
    launch();
    CK(hipStreamSynchronize(st));
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    CK(hipEventRecord(t0, st));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(t1, st));
    CK(hipEventSynchronize(t1));
    float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
    ms /= iters;
    CK(hipGetLastError());
    printf("M=%d N=%d K=%d ld=%d batch=%d%s%s: %.3f ms per launch, %.1f TFLOP/s (8MNK)\n", M, N, K, ld, batch, zero ? " zero" : "", shared ? " shared" : "",
           ms, 8.0 * M * N * K * batch / (ms * 1e-3) / 1e12);
    (void)hipFree(base);
    return 0;
}
