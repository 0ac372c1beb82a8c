// Legacy NumPy MT19937 stream helpers (host).  maus_mt19937_jump advances a RandomState key/pos
// by `nwords` 32-bit outputs.  (Interim implementation: steps the generator; the GF(2) jump
// polynomial replaces the loop in the RNG milestone.)
#include <stdint.h>
#include "../../include/maus_hip.h"

namespace {
constexpr int N = 624, M = 397;
inline void regen(uint32_t* mt) {
    const uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MA = 0x9908b0dfu;
    int k = 0; uint32_t y;
    for (; k < N - M; ++k) { y = (mt[k] & UP) | (mt[k + 1] & LO); mt[k] = mt[k + M] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u); }
    for (; k < N - 1; ++k) { y = (mt[k] & UP) | (mt[k + 1] & LO); mt[k] = mt[k + (M - N)] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u); }
    y = (mt[N - 1] & UP) | (mt[0] & LO); mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
}
}  // namespace

extern "C" int maus_mt19937_jump(uint32_t* key, int32_t* pos, uint64_t nwords) {
    if (!key || !pos || *pos < 0 || *pos > N) return -1;
    uint64_t p = (uint64_t)*pos + nwords;       // outputs are consumed from key[pos]; pos==624 triggers a regen first
    while (p > (uint64_t)N) { regen(key); p -= N; }
    *pos = (int32_t)p;
    return 0;
}
