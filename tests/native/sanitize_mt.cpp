// Host-only driver for the CPU sanitizer build (`make -C adaptive_matrix_solver_amd/csrc asan`): exercises the legacy
// NumPy MT19937 stream code that the product links -- the GF(2) jump (mt19937.cpp) and the regeneration plan of a device
// sub-batch (mtplan.cpp) -- against plain block-by-block stepping.  No HIP, no GPU.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../adaptive_matrix_solver_amd/csrc/mtplan.h"

int maus_mt_jump_poly(uint64_t J, uint64_t* out312);

namespace {
constexpr int N = 624, M = 397;
constexpr uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MA = 0x9908b0dfu;

void regen(uint32_t* mt) {
    int k = 0; uint32_t y;
    for (; k < N - M; ++k) { y = (mt[k] & UP) | (mt[k + 1] & LO); mt[k] = mt[k + M] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u); }
    for (; k < N - 1; ++k) { y = (mt[k] & UP) | (mt[k + 1] & LO); mt[k] = mt[k + (M - N)] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u); }
    y = (mt[N - 1] & UP) | (mt[0] & LO); mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
}
void seed(uint32_t* mt, uint32_t s) { mt[0] = s; for (int i = 1; i < N; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i; }

// NumPy's consumption: word at `pos`; pos == 624 -> regenerate first
void step_words(uint32_t* key, int& pos, uint64_t nwords) {
    for (uint64_t i = 0; i < nwords; ++i) { if (pos == N) { regen(key); pos = 0; } ++pos; }
}
int fails = 0;
void check(bool ok, const char* what) { if (!ok) { std::printf("FAIL: %s\n", what); ++fails; } }
}  // namespace

int main() {
    // (1) the jump against real stepping, from several positions, over sub-block, block-straddling and long distances
    const uint64_t dist[] = {0, 1, 7, 623, 624, 625, 4096, 100003, 4ull * 64 * 64, 4ull * 300 * 300};
    const int starts[] = {624, 1, 5, 333, 623};
    for (int sp : starts)
        for (uint64_t J : dist) {
            uint32_t a[N], b[N]; seed(a, 5489u + (uint32_t)sp); regen(a); std::memcpy(b, a, sizeof a);
            int pa = sp; int32_t pb = sp;
            step_words(a, pa, J);
            check(maus_mt19937_jump(b, &pb, J) == 0, "maus_mt19937_jump returned an error");
            // NumPy regenerates lazily: a state with pos == 624 is the same stream point as the next block at pos 0
            if (pa != pb) { check(false, "jump: position differs"); continue; }
            check(std::memcmp(a, b, sizeof a) == 0, "jump: key differs from stepping");
        }
    // (2) the plan of a device sub-batch: every generator's start, reached by the planned jumps + `extra` regenerations,
    //     must be the state plain stepping reaches at that generator's first word
    struct Cfg { int n, g, first, pos, s_override; uint64_t wpc_mult, lead_mult; };
    const Cfg cfgs[] = {{24, 5, 0, 624, 0, 2, 0}, {40, 9, 2, 17, 3, 4, 2}, {64, 33, 0, 600, 0, 2, 0}, {96, 3, 1, 1, 4, 2, 0},
                        {96, 4, 0, 600, 7, 2, 0}, {48, 3, 1, 5, 16, 2, 1}, {128, 3, 0, 33, 13, 2, 0}, {128, 2, 0, 624, 16, 2, 0}};
    for (const Cfg& c : cfgs) {
        maus_mt_desc d; std::memset(&d, 0, sizeof d);
        uint32_t base[N]; seed(base, 12345u + (uint32_t)c.n); regen(base);
        std::memcpy(d.key, base, sizeof base);
        const uint64_t two_n2 = 2ull * c.n * c.n;
        d.pos = c.pos; d.words_per_candidate = c.wpc_mult * two_n2; d.lead_words = c.lead_mult * two_n2;
        std::vector<int32_t> ords(c.first + c.g);
        for (size_t i = 0; i < ords.size(); ++i) ords[i] = (int32_t)(i * 2 + (i % 3));           // ragged, increasing
        d.ordinals = ords.data();
        MausMtPlan pl; const char* err = nullptr;
        check(maus_mt_plan(&d, c.n, c.first, c.g, c.s_override, &pl, &err) == 0, err ? err : "maus_mt_plan failed");
        check(pl.ngen == 2 * c.g * pl.S && (int)pl.hs.size() >= 2 * pl.ngen, "plan: sizes");
        // blocks each generator is advanced by the levels, in order: lifting over the draw index in place (src_off = 0),
        // then the doubling tree over the sub-stream index (state <- jump of the state src_off generators before it)
        std::vector<uint64_t> jumped(pl.ngen, 0);
        for (const auto& L : pl.levels) {
            check(L.J % 624 == 0 && L.off + (size_t)L.count <= pl.hs.size(), "plan: level bounds");
            check(L.off + (size_t)L.count * (L.multi ? 2 : 1) <= pl.hs.size(), "plan: level bounds (multipliers)");
            for (int i = 0; i < L.count; ++i) {
                const int gi = pl.hs[L.off + i];
                const int mult = L.multi ? pl.hs[L.off + L.count + i] : 1;
                check(gi >= 0 && gi < pl.ngen && gi - L.src_off >= 0 && mult >= 1 && mult <= 15, "plan: generator index / multiplier out of range");
                if (gi >= 0 && gi < pl.ngen && gi - L.src_off >= 0) jumped[gi] = jumped[gi - L.src_off] + (uint64_t)mult * (L.J / 624);
            }
        }
        for (int k = 0; k < c.g; ++k)
            for (int sb = 0; sb < pl.S; ++sb)
                for (int part = 0; part < 2; ++part) {
                    const int gi = (k * pl.S + sb) * 2 + part;
                    const uint64_t mm = (d.lead_words + (uint64_t)ords[c.first + k] * d.words_per_candidate) / two_n2 + part;
                    const uint64_t t = (uint64_t)d.pos + mm * two_n2 + 2ull * sb * pl.E;
                    check(jumped[gi] + (uint64_t)pl.hs[gi] == t / 624 && (uint64_t)pl.hs[pl.ngen + gi] == t % 624, "plan: block arithmetic");
                    if (k < 2 && (sb == pl.S - 1 || sb == pl.S / 2)) {   // and really: jump + regenerate == step
                        uint32_t a[N], b[N]; std::memcpy(a, base, sizeof a); std::memcpy(b, base, sizeof b);
                        for (uint64_t q = 0; q < t / 624; ++q) regen(a);
                        int32_t pb = 624;                             // key as a block boundary: J words = J/624 blocks
                        if (jumped[gi]) check(maus_mt19937_jump(b, &pb, jumped[gi] * 624) == 0 && pb == 624, "plan: jump");
                        for (int q = 0; q < pl.hs[gi]; ++q) regen(b);
                        check(std::memcmp(a, b, sizeof a) == 0, "plan: generator start state differs from stepping");
                    }
                }
    }
    // (3) error paths
    { maus_mt_desc d; std::memset(&d, 0, sizeof d); MausMtPlan pl; const char* err = nullptr; int32_t o = 0; d.ordinals = &o;
      d.pos = 700; d.words_per_candidate = 8; check(maus_mt_plan(&d, 1, 0, 1, 0, &pl, &err) != 0 && err, "plan: bad pos accepted");
      d.pos = 3; d.words_per_candidate = 3; check(maus_mt_plan(&d, 2, 0, 1, 0, &pl, &err) != 0, "plan: bad wpc accepted"); }
    uint64_t poly[312];
    check(maus_mt_jump_poly(624, poly) == 0, "jump polynomial");
    std::printf(fails ? "sanitize_mt: %d check(s) failed\n" : "sanitize_mt: all checks passed\n", fails);
    return fails ? 1 : 0;
}
