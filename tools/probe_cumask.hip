// Which CUs does a CU-masked stream run on?  (hipExtStreamCreateWithCUMask on gfx950: 8 XCDs x 32 CUs)
// Each workgroup records (XCC_ID, SE_ID, CU_ID) from the hardware-id registers; the host prints how many
// distinct CUs per XCD were used under a few masks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <set>

__global__ void where_kernel(unsigned* out, int spin) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(4)" : "=s"(hwid));     // HW_REG_HW_ID
    asm volatile("s_getreg_b32 %0, hwreg(20)" : "=s"(xcc));     // HW_REG_XCC_ID
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 0xf) << 16 | (hwid & 0xffff);
    // keep the workgroup resident for a while so that the grid spreads over every allowed CU
    long long t0 = clock64();
    while (clock64() - t0 < spin) { }
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", name, hipGetErrorString(e)); return; }
    const int nwg = 4096;
    unsigned* d; hipMalloc(&d, nwg * sizeof(unsigned));
    hipLaunchKernelGGL(where_kernel, dim3(nwg), dim3(256), 0, st, d, 200000);
    hipStreamSynchronize(st);
    std::vector<unsigned> h(nwg);
    hipMemcpy(h.data(), d, nwg * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::set<unsigned> per[16];
    for (unsigned v : h) { unsigned x = v >> 16, se = (v >> 13) & 7, cu = (v >> 8) & 15; per[x & 15].insert(se << 4 | cu); }
    printf("%-28s CUs per XCC:", name);
    int tot = 0;
    for (int x = 0; x < 8; ++x) { printf(" %2zu", per[x].size()); tot += (int)per[x].size(); }
    printf("  total %d\n", tot);
    hipFree(d); hipStreamDestroy(st);
}

int main() {
    std::vector<uint32_t> full(8, 0xffffffffu);
    run("all 256 bits", full);
    std::vector<uint32_t> lo(8, 0); for (int i = 0; i < 4; ++i) lo[i] = 0xffffffffu;
    run("bits 0..127", lo);
    std::vector<uint32_t> hi(8, 0); for (int i = 4; i < 8; ++i) hi[i] = 0xffffffffu;
    run("bits 128..255", hi);
    std::vector<uint32_t> even(8, 0x55555555u);
    run("even bits", even);
    std::vector<uint32_t> q(8, 0); q[0] = 0xffffffffu; q[1] = 0xffffffffu;
    run("bits 0..63", q);
    std::vector<uint32_t> m8(8, 0x000000ffu);
    run("low byte of every word", m8);
    std::vector<uint32_t> w0(8, 0); w0[0] = 0xffffffffu;
    run("bits 0..31", w0);
    std::vector<uint32_t> e8(8, 0x01010101u);
    run("every 8th bit (i%8==0)", e8);
    std::vector<uint32_t> p32(8, 0); for (int i = 0; i < 256; ++i) if ((i / 8) % 8 == 0) p32[i / 32] |= 1u << (i % 32);
    run("bits with (i/8)%8==0", p32);
    std::vector<uint32_t> c32(8, 0); for (int i = 0; i < 256; ++i) if ((i / 8) % 8 != 0) c32[i / 32] |= 1u << (i % 32);
    run("complement of that", c32);
    return 0;
}
