// Legacy NumPy MT19937 stream helpers (host side).
//
// The reference draws 2 x np.random.rand(N,N) per dense solve attempt (AMS:49) from the global
// legacy RandomState: 4*N*N MT19937 words whose VALUES are numerically inert at the default
// psi but whose CONSUMPTION positions every later draw (candidate re-inits,
// spawns).  maus_mt19937_jump advances a (key[624], pos) pair by `nwords` outputs without
// generating them: state(t+J) = g_J(F) state(t) with g_J(x) = x^J mod phi(x) over GF(2),
// phi = characteristic polynomial of the MT19937 transition F (degree 19937), evaluated by
// Horner's rule (Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer 2008).  phi is obtained
// once by Berlekamp-Massey on the generator's own output bits; g_J is cached per J.
//
// The result is bit-identical to the state NumPy reaches by actually drawing the words,
// including its lazy block regeneration (pos stays in 1..624).
#include <stdint.h>
#include <string.h>
#include <map>
#include <mutex>
#include <vector>
#include "../../include/maus_hip.h"

namespace {

constexpr int N = 624, M = 397;
constexpr int DEG = 19937;
constexpr int PW = (DEG + 64) / 64;          // 312 words hold degrees 0..19967
constexpr uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MA = 0x9908b0dfu;

inline void regen(uint32_t* mt) {             // next block of 624 words, in place (time-ordered)
    int k = 0; uint32_t y;
    for (; k < N - M; ++k) { y = (mt[k] & UP) | (mt[k + 1] & LO); mt[k] = mt[k + M] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u); }
    for (; k < N - 1; ++k) { y = (mt[k] & UP) | (mt[k + 1] & LO); mt[k] = mt[k + (M - N)] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u); }
    y = (mt[N - 1] & UP) | (mt[0] & LO); mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
}

struct Poly { uint64_t w[PW]; };              // bit i = coefficient of x^i, degree < 19968
inline bool pbit(const Poly& p, int i) { return (p.w[i >> 6] >> (i & 63)) & 1ull; }
inline void pflip(Poly& p, int i) { p.w[i >> 6] ^= 1ull << (i & 63); }

Poly g_phi;                                   // phi(x) including the leading x^19937
bool g_phi_ok = false;
std::mutex g_mu;
std::map<uint64_t, Poly> g_cache;             // J -> x^J mod phi

// ---- Berlekamp-Massey over GF(2), bit-packed ------------------------------------------------
// sequence s_0..s_{L-1}; returns connection polynomial C (C_0 = 1) of the shortest LFSR:
//   sum_{i=0..deg} C_i s_{n-i} = 0.
void berlekamp_massey(const std::vector<uint8_t>& s, std::vector<uint64_t>& C, int& Lout) {
    const int n = (int)s.size();
    const int W = n / 64 + 2;
    std::vector<uint64_t> Cc(W, 0), B(W, 0), T(W, 0);
    // reversed sequence for windowed dot products: R bit j = s_{n-1-j}
    std::vector<uint64_t> R(W + 1, 0);
    for (int j = 0; j < n; ++j) if (s[n - 1 - j]) R[j >> 6] |= 1ull << (j & 63);
    Cc[0] = 1; B[0] = 1;
    int L = 0, m = 1;
    for (int i = 0; i < n; ++i) {
        // d = sum_{k=0..L} C_k s_{i-k};  s_{i-k} = R bit (n-1-i+k)
        const int off = n - 1 - i;
        const int wo = off >> 6, bo = off & 63;
        uint64_t acc = 0;
        const int words = L / 64 + 1;
        for (int k = 0; k < words; ++k) {
            uint64_t win = R[wo + k] >> bo;
            if (bo) win |= R[wo + k + 1] << (64 - bo);
            acc ^= Cc[k] & win;
        }
        // mask off coefficients above L in the last word
        // (coefficients above L are zero by construction, so no mask is needed)
        const int d = __builtin_parityll(acc);
        if (d == 0) { ++m; continue; }
        if (2 * L <= i) {
            T = Cc;
            // C ^= B << m
            const int ws = m >> 6, bs = m & 63;
            for (int k = W - 1; k >= ws; --k) {
                uint64_t v = B[k - ws] << bs;
                if (bs && k - ws - 1 >= 0) v |= B[k - ws - 1] >> (64 - bs);
                Cc[k] ^= v;
            }
            L = i + 1 - L; B = T; m = 1;
        } else {
            const int ws = m >> 6, bs = m & 63;
            for (int k = W - 1; k >= ws; --k) {
                uint64_t v = B[k - ws] << bs;
                if (bs && k - ws - 1 >= 0) v |= B[k - ws - 1] >> (64 - bs);
                Cc[k] ^= v;
            }
            ++m;
        }
    }
    C = Cc; Lout = L;
}

bool init_phi() {
    // output bits of a fixed, well-mixed state: lsb of the untempered words
    uint32_t mt[N];
    mt[0] = 19650218u;
    for (int i = 1; i < N; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    for (int k = 0; k < 4; ++k) regen(mt);
    const int need = 2 * DEG + 128;
    std::vector<uint8_t> s(need);
    int p = N;
    for (int i = 0; i < need; ++i) {
        if (p >= N) { regen(mt); p = 0; }
        s[i] = (uint8_t)(mt[p++] & 1u);
    }
    std::vector<uint64_t> C; int L = 0;
    berlekamp_massey(s, C, L);
    if (L != DEG) return false;
    // characteristic polynomial = reciprocal of the connection polynomial: phi_i = C_{DEG-i}
    memset(&g_phi, 0, sizeof g_phi);
    for (int i = 0; i <= DEG; ++i) if ((C[i >> 6] >> (i & 63)) & 1ull) pflip(g_phi, DEG - i);
    return pbit(g_phi, DEG) && pbit(g_phi, 0);
}

// r = a mod phi, a given with up to 2*PW words
void reduce(std::vector<uint64_t>& a, Poly& r) {
    const int top = (int)a.size() * 64 - 1;
    for (int i = top; i >= DEG; --i) {
        if (!((a[i >> 6] >> (i & 63)) & 1ull)) continue;
        const int sh = i - DEG, ws = sh >> 6, bs = sh & 63;
        for (int k = 0; k < PW; ++k) {
            uint64_t v = g_phi.w[k];
            a[k + ws] ^= v << bs;
            if (bs) a[k + ws + 1] ^= v >> (64 - bs);
        }
    }
    memcpy(r.w, a.data(), sizeof(uint64_t) * PW);
    // clear anything at or above DEG (all zero after reduction)
}

// ---- Barrett reduction with carry-less multiplies (x86 PCLMULQDQ), round 3 -----------------------------------------------
// The schoolbook reduce() above costs ~10 000 x 312 word XORs per call and x^J mod phi needs one per bit of J and more: 25-85 ms
// per new J -- and the engine asks for a new J every loop body (the whole run's E3 consumption: 4 N^2 words x candidates), which
// made 26 ms the floor of a GMRES loop body whose kernels take 12.  With mu = floor(x^(2 DEG) / phi) precomputed,
//     q = floor( floor(a / x^DEG) * mu / x^DEG ),   a mod phi = (a xor q * phi) mod x^DEG          (deg a < 2 DEG; exact over GF(2))
// is two products of 312-word polynomials: ~0.2 ms.  Falls back to reduce() on a CPU without PCLMULQDQ.
#if defined(__x86_64__) && defined(__PCLMUL__)
#include <smmintrin.h>
#include <wmmintrin.h>
#define MAUS_HAVE_CLMUL 1
Poly g_barrett;                               // mu = floor(x^(2 DEG) / phi), degree DEG
bool g_barrett_ok = false;

// c[0 .. na+nb) = a * b over GF(2); if low_only > 0 only words below low_only are produced
void pmul(const uint64_t* a, int na, const uint64_t* b, int nb, uint64_t* c, int low_only = 0) {
    const int nc = low_only > 0 ? low_only : na + nb;
    for (int k = 0; k < nc; ++k) c[k] = 0;
    for (int i = 0; i < na; ++i) {
        if (!a[i]) continue;
        const __m128i ai = _mm_set_epi64x(0, (long long)a[i]);
        const int jmax = low_only > 0 ? (low_only - i < nb ? low_only - i : nb) : nb;
        for (int j = 0; j < jmax; ++j) {
            const __m128i p = _mm_clmulepi64_si128(ai, _mm_set_epi64x(0, (long long)b[j]), 0x00);
            c[i + j] ^= (uint64_t)_mm_cvtsi128_si64(p);
            if (i + j + 1 < nc) c[i + j + 1] ^= (uint64_t)_mm_extract_epi64(p, 1);
        }
    }
}

// out[0 .. nout) = a[0 .. na) >> sh bits
void pshr(const uint64_t* a, int na, int sh, uint64_t* out, int nout) {
    const int ws = sh >> 6, bs = sh & 63;
    for (int k = 0; k < nout; ++k) {
        uint64_t v = (k + ws < na) ? a[k + ws] >> bs : 0;
        if (bs && k + ws + 1 < na) v |= a[k + ws + 1] << (64 - bs);
        out[k] = v;
    }
}

void init_mu() {                              // long division of x^(2 DEG) by phi, once
    std::vector<uint64_t> rem(2 * PW + 2, 0);
    memset(&g_barrett, 0, sizeof g_barrett);
    const int top = 2 * DEG;
    rem[top >> 6] |= 1ull << (top & 63);
    for (int i = top; i >= DEG; --i) {
        if (!((rem[i >> 6] >> (i & 63)) & 1ull)) continue;
        const int sh = i - DEG, ws = sh >> 6, bs = sh & 63;
        pflip(g_barrett, sh);
        for (int k = 0; k < PW; ++k) {
            const uint64_t v = g_phi.w[k];
            rem[k + ws] ^= v << bs;
            if (bs) rem[k + ws + 1] ^= v >> (64 - bs);
        }
    }
}

void reduce_clmul(const std::vector<uint64_t>& a, Poly& r) {      // a: 2 PW + 2 words, degree < 2 DEG
    uint64_t a1[PW + 1], t[2 * PW + 2], q[PW + 1], qp[PW + 1];
    pshr(a.data(), (int)a.size(), DEG, a1, PW + 1);               // floor(a / x^DEG): degree < DEG
    pmul(a1, PW, g_barrett.w, PW, t);                                    // (a1 has no bits at or above DEG: PW words suffice)
    t[2 * PW] = t[2 * PW + 1] = 0;
    pshr(t, 2 * PW + 2, DEG, q, PW + 1);                            // q = floor(a1 mu / x^DEG)
    pmul(q, PW, g_phi.w, PW, qp, PW);                               // low PW words of q * phi
    for (int k = 0; k < PW; ++k) r.w[k] = a[k] ^ qp[k];
    r.w[PW - 1] &= (DEG & 63) ? ((1ull << (DEG & 63)) - 1ull) : ~0ull;      // mod x^DEG (DEG = 311 * 64 + 33)
}
#endif

inline uint64_t spread32(uint32_t x) {       // interleave zeros: bit i -> bit 2i
    uint64_t v = x;
    v = (v | (v << 16)) & 0x0000FFFF0000FFFFull;
    v = (v | (v << 8)) & 0x00FF00FF00FF00FFull;
    v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
}

#ifdef MAUS_HAVE_CLMUL
bool use_clmul() { static const bool ok = __builtin_cpu_supports("pclmul"); return ok && g_barrett_ok; }
#define REDUCE(tmp, r) do { if (use_clmul()) reduce_clmul(tmp, r); else reduce(tmp, r); } while (0)
#else
#define REDUCE(tmp, r) reduce(tmp, r)
#endif

void poly_pow_x(uint64_t J, Poly& out) {      // x^J mod phi, square-and-multiply on the bits of J
    Poly r; memset(&r, 0, sizeof r); r.w[0] = 1;               // 1
    std::vector<uint64_t> tmp(2 * PW + 2);
    int top = 63; while (top > 0 && !((J >> top) & 1ull)) --top;
    for (int b = top; b >= 0; --b) {
        // square
        std::fill(tmp.begin(), tmp.end(), 0);
        for (int k = 0; k < PW; ++k) { tmp[2 * k] = spread32((uint32_t)r.w[k]); tmp[2 * k + 1] = spread32((uint32_t)(r.w[k] >> 32)); }
        REDUCE(tmp, r);
        if ((J >> b) & 1ull) {                                   // times x
            std::fill(tmp.begin(), tmp.end(), 0);
            uint64_t carry = 0;
            for (int k = 0; k < PW; ++k) { tmp[k] = (r.w[k] << 1) | carry; carry = r.w[k] >> 63; }
            tmp[PW] = carry;
            REDUCE(tmp, r);
        }
    }
    out = r;
}

// s (time-ordered 624 words) <- F^J s via Horner with g = x^J mod phi
void apply_poly(const Poly& g, uint32_t* s) {
    uint32_t buf[N]; memset(buf, 0, sizeof buf);
    int h = 0; bool started = false;
    for (int i = DEG - 1; i >= 0; --i) {
        if (started) {                      // acc <- F(acc)
            uint32_t y = (buf[h] & UP) | (buf[h + 1 == N ? 0 : h + 1] & LO);
            int hm = h + M; if (hm >= N) hm -= N;
            buf[h] = buf[hm] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
            if (++h == N) h = 0;
        }
        if (pbit(g, i)) {                   // acc ^= s   (element k of the state lives at buf[(h+k)%N])
            started = true;
            const int first = N - h;
            for (int k = 0; k < first; ++k) buf[h + k] ^= s[k];
            for (int k = first; k < N; ++k) buf[k - first] ^= s[k];
        }
    }
    for (int k = 0; k < N; ++k) { int idx = h + k; if (idx >= N) idx -= N; s[k] = buf[idx]; }
}

// g = x^J mod phi through the cache.  The mutex covers only the lookups: a new polynomial takes tens of
// milliseconds, and the engine's worker thread (whole-run advance) and the device path (lifting polynomials)
// ask for different J concurrently.
int cached_poly(uint64_t J, Poly& g) {
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_phi_ok) { if (!init_phi()) return -2; g_phi_ok = true; }
#ifdef MAUS_HAVE_CLMUL
        if (!g_barrett_ok) { init_mu(); g_barrett_ok = true; }
#endif
        auto it = g_cache.find(J);
        if (it != g_cache.end()) { g = it->second; return 0; }
    }
    poly_pow_x(J, g);
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_cache.size() > 256) g_cache.clear();
    g_cache[J] = g;
    return 0;
}

}  // namespace

extern "C" int maus_mt19937_jump(uint32_t* key, int32_t* pos, uint64_t nwords) {
    if (!key || !pos || *pos < 0 || *pos > N) return -1;
    if (nwords == 0) return 0;
    const uint64_t total = (uint64_t)*pos + nwords;      // >= 1
    const uint64_t qb = (total - 1) / N;                 // number of block regenerations NumPy would perform
    const int32_t newpos = (int32_t)(total - qb * N);    // 1..624
    if (qb == 0) { *pos = newpos; return 0; }
    if (qb <= 4) { for (uint64_t k = 0; k < qb; ++k) regen(key); *pos = newpos; return 0; }
    // F^(624*(qb-1)) by the polynomial, then one real regeneration (which also repairs the 31
    // low bits of word 0 that the 19937-bit state does not carry)
    const uint64_t J = (uint64_t)N * (qb - 1);
    Poly g;
    if (cached_poly(J, g)) return -2;
    apply_poly(g, key);
    regen(key);
    *pos = newpos;
    return 0;
}

// x^J mod phi as 312 little-endian 64-bit words (bit i = coefficient of x^i); cached.  Used by the device-side
// stream regeneration (mtdev.hip) for its binary-lifting jumps.
int maus_mt_jump_poly(uint64_t J, uint64_t* out312) {
    Poly g;
    if (cached_poly(J, g)) return -2;
    memcpy(out312, g.w, sizeof(uint64_t) * PW);
    return 0;
}
