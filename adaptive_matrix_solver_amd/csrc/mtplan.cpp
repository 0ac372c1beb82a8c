// See mtplan.h.  Candidate k of a run uses the 4N^2 words at offset lead + ord_k * words_per_candidate of the current
// NumPy stream (AMS:49: two rand(N,N) per dense attempt); each of its two draws is cut into S sub-streams of
// ceil(N^2/S) elements.  Generator (k, sb, part) therefore starts at stream word
//     t = pos + (draw index mm) * 2N^2 + 2 * sb * E,         mm = (lead + ord_k * wpc) / 2N^2 + part,
// i.e. in block q = t / 624 at position t % 624.  q is reached as  m * dj + b * dj2  blocks by jump polynomials (binary
// lifting: level i applies x^(624 * stride * 2^i) to the states whose index has bit i set) plus `extra` real block
// regenerations inside the build kernel.
#include "mtplan.h"
#include <algorithm>

int maus_mt_plan(const maus_mt_desc* d, int n, int first, int g, int s_override, MausMtPlan* out, const char** err) {
    auto fail = [&](const char* m) { if (err) *err = m; return -1; };
    if (!d || !out || n <= 0 || g <= 0 || first < 0) return fail("maus_mt_plan: bad arguments");
    const uint64_t two_n2 = 2ull * n * n;
    if (d->pos < 0 || d->pos > 624) return fail("maus_mt_desc: bad position");
    if (d->words_per_candidate % two_n2 || d->lead_words % two_n2 || d->words_per_candidate < 2 * two_n2 || !d->ordinals)
        return fail("maus_mt_desc: words_per_candidate / lead_words must be multiples of 2*n*n");
    // sub-streams per draw: enough workgroups to cover the chip a few times, each at least ~64 blocks long
    int S = std::max(1, std::min(8, 768 / std::max(1, g)));
    const uint64_t nn = (uint64_t)n * n;
    while (S > 1 && nn / S < 64 * 312) --S;
    if (s_override > 0) S = std::max(1, std::min(16, s_override));
    const uint64_t E = (nn + S - 1) / S;                          // elements per sub-stream
    const int ngen = 2 * g * S;
    const uint64_t dblocks = two_n2 / 624;
    const uint64_t dj = dblocks >= 2 ? dblocks - 1 : 0;          // jump stride (blocks) per draw; >= 1 real regeneration follows
    const uint64_t sblocks = (2 * E) / 624;
    const uint64_t dj2 = (S > 1 && sblocks >= 2) ? sblocks - 1 : 0;   // jump stride (blocks) per sub-stream
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
    auto plan = [&](const std::vector<uint64_t>& idx, uint64_t maxv, uint64_t stride_blocks) {
        for (int bit = 0; stride_blocks && bit < 64 && (maxv >> bit); ++bit) {
            const size_t off = hs.size();
            for (int i = 0; i < ngen; ++i) if ((idx[i] >> bit) & 1ull) hs.push_back(i);
            const int cnt = (int)(hs.size() - off);
            if (!cnt) continue;
            out->levels.push_back({off, cnt, 624ull * stride_blocks * (1ull << bit)});
        }
    };
    plan(m, maxm, dj);
    plan(bsel, maxb, dj2);
    out->S = S; out->E = E; out->ngen = ngen; out->dj = dj; out->dj2 = dj2;
    return 0;
}
