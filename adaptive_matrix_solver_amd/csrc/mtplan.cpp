// See mtplan.h.  Candidate k of a run uses the 4N^2 words at offset lead + ord_k * words_per_candidate of the current
// NumPy stream (AMS:49: two rand(N,N) per dense attempt); each of its two draws is cut into S sub-streams of
// ceil(N^2/S) elements.  Generator (k, sb, part) therefore starts at stream word
//     t = pos + (draw index mm) * 2N^2 + 2 * sb * E,         mm = (lead + ord_k * wpc) / 2N^2 + part,
// i.e. in block q = t / 624 at position t % 624.  q is reached as  m * dj + b * dj2  blocks by jump polynomials plus
// `extra` real block regenerations inside the build kernel:
//   * over the draw index m by lifting in base 16 (level i applies x^(v * 624 * dj * 16^i) to the states whose m has the
//     hexadecimal digit v at position i), on the first sub-stream of every draw only;
//   * over the sub-stream index b by a doubling tree: level i computes the states with b in [2^i, 2^(i+1)) from the
//     states with b - 2^i (one jump per generator; lifting every sub-stream on its own cost popcount(m) + popcount(b)
//     jumps each, 13 launches of ~2900 jump workgroups = 22 ms at 181 candidates x 16 sub-streams).
// dj and dj2 are the whole blocks a draw / a sub-stream is long, so `extra` stays a handful of blocks (round 2 jumped one
// block short per step "so that a real regeneration follows": word 0 of a jumped state has correct low bits only if the
// base state has a predecessor; a freshly seeded state (pos = 624) has not, but then t / 624 exceeds the jumped blocks
// by at least one anyway, and every state NumPy reaches with pos < 624 has been regenerated at least once).
#include "mtplan.h"
#include <algorithm>

int maus_mt_plan(const maus_mt_desc* d, int n, int first, int g, int s_override, MausMtPlan* out, const char** err) {
    auto fail = [&](const char* m) { if (err) *err = m; return -1; };
    if (!d || !out || n <= 0 || g <= 0 || first < 0) return fail("maus_mt_plan: bad arguments");
    const uint64_t two_n2 = 2ull * n * n;
    if (d->pos < 0 || d->pos > 624) return fail("maus_mt_desc: bad position");
    if (d->words_per_candidate % two_n2 || d->lead_words % two_n2 || d->words_per_candidate < 2 * two_n2 || !d->ordinals)
        return fail("maus_mt_desc: words_per_candidate / lead_words must be multiples of 2*n*n");
    // sub-streams per draw (one 4-wave workgroup each): a power of two, the smallest that gives the H build ~1000 workgroups
    // (measured at 181 candidates, n = 4096: 8 -> 15.3 ms, 9 -> 18.5, 16 -> 16.5, 4 -> 21.9 per H build).  Not more: every
    // further sub-stream costs a jump, i.e. as much LDS traffic as generating 3300 blocks -- at n = 1024 a whole draw -- and
    // at 256 candidates of n = 1024 eight sub-streams instead of four cost 2048 more jumps (0.96 ms) for the same 1.7 ms of
    // H build (round 4: 21.0 -> 20.1 ms per 256-solve call).  Each sub-stream at least ~64 blocks long.
    // Small batches of large matrices go on to 64 sub-streams (32 candidates at n = 4096: 32 instead of 16 -> H build 4.9 -> 3.4 ms,
    // the call 109.0 -> 107.5; at n = 1024 the extra jumps would cost more than the build gains).
    const int s_cap = (n >= 2048) ? 64 : 16;
    int S = 1;
    while (2 * S <= s_cap && (long)S * std::max(1, g) < 1024) S *= 2;
    const uint64_t nn = (uint64_t)n * n;
    while (S > 1 && nn / S < 64 * 312) --S;
    if (s_override > 0) S = std::max(1, std::min(64, s_override));
    const uint64_t E = (nn + S - 1) / S;                          // elements per sub-stream
    const int ngen = 2 * g * S;
    const uint64_t dblocks = two_n2 / 624;
    const uint64_t dj = dblocks;                                 // jump stride (blocks) per draw
    const uint64_t sblocks = (2 * E) / 624;
    const uint64_t dj2 = (S > 1) ? sblocks : 0;                  // jump stride (blocks) per sub-stream
    std::vector<uint64_t> m(ngen), bsel(ngen);
    std::vector<int>& hs = out->hs;
    hs.assign(2 * (size_t)ngen, 0);
    uint64_t maxm = 0, maxb = 0;
    for (int k = 0; k < g; ++k) {
        if (d->ordinals[first + k] < 0) return fail("maus_mt_desc: negative ordinal");
        const uint64_t ord = (uint64_t)d->ordinals[first + k];
        for (int sb = 0; sb < S; ++sb)
            for (int part = 0; part < 2; ++part) {
                const int gi = (k * S + sb) * 2 + part;
                const uint64_t mm = (d->lead_words + ord * d->words_per_candidate) / two_n2 + part;
                const uint64_t t = (uint64_t)d->pos + mm * two_n2 + 2ull * sb * E;
                const uint64_t q = t / 624;
                m[gi] = dj ? mm : 0;
                bsel[gi] = dj2 ? (uint64_t)sb : 0;
                const uint64_t ex = q - m[gi] * dj - bsel[gi] * dj2;
                if (ex > 2000000000ull) return fail("maus_mt_desc: stream offset too large");
                hs[gi] = (int)ex;
                hs[ngen + gi] = (int)(t % 624);
                maxm = std::max(maxm, m[gi]); maxb = std::max(maxb, bsel[gi]);
            }
    }
    out->levels.clear();
    // lifting over m, one HEXADECIMAL digit per level (a launch applies x^(v J) with the generator's own digit v: three
    // dependent launches for m < 4096 where binary lifting needed twelve, each ~0.2-0.35 ms whatever it carries): on the first
    // sub-stream of each draw when the tree below derives the others from it, else on all
    for (int dig = 0; dj && 4 * dig < 64 && (maxm >> (4 * dig)); ++dig) {
        const size_t off = hs.size();
        std::vector<int> mult;
        for (int i = 0; i < ngen; ++i) {
            const int v = (int)((m[i] >> (4 * dig)) & 15ull);
            if (v && (!dj2 || (i / 2) % S == 0)) { hs.push_back(i); mult.push_back(v); }
        }
        const int cnt = (int)mult.size();
        hs.insert(hs.end(), mult.begin(), mult.end());
        if (cnt) out->levels.push_back({off, cnt, 624ull * dj << (4 * dig), 0, true});
        else hs.resize(off);
    }
    // doubling tree over the sub-stream index: state(b) = x^(624 * dj2 * 2^i) state(b - 2^i) for 2^i <= b < 2^(i+1)
    for (int bit = 0; dj2 && (1 << bit) < S; ++bit) {
        const size_t off = hs.size();
        for (int i = 0; i < ngen; ++i) { const int sb = (i / 2) % S; if (sb >= (1 << bit) && sb < (2 << bit)) hs.push_back(i); }
        const int cnt = (int)(hs.size() - off);
        if (cnt) out->levels.push_back({off, cnt, 624ull * dj2 * (1ull << bit), 2 << bit, false});
    }
    (void)maxb;
    out->S = S; out->E = E; out->ngen = ngen; out->dj = dj; out->dj2 = dj2;
    return 0;
}
