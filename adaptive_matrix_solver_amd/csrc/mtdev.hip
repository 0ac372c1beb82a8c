// Device-side regeneration of the legacy NumPy MT19937 stream for the dense regulariser (AMS:49-50,
// SURVEY F4 / §8f row f-1):  reg = psi*I + 0.15*psi*((U1-.5) + i(U2-.5)),  U1 then U2 = np.random.rand(N,N),
// i.e. 4*N*N consecutive MT19937 words per solve attempt.  With MAUS_PERT_MT19937 the H_k are built
// from bit-identical draws without the host ever materialising them:
//
//  1. every candidate needs two generator start states (U1 and U2), all at offsets m * 2N^2 words from
//     the current NumPy state.  They are reached by binary lifting over m with jump polynomials
//     g_i(x) = x^(624*Dj*2^i) mod phi (host, cached; Dj = floor(2N^2/624) - 1 blocks).  A jump is
//     evaluated as a CONVOLUTION: g(F) s = sum_i g_i F^i s and F^i s is the stream shifted by i words,
//     so out[j] = XOR_{i : g_i = 1} x[i + j] over the next 19937+624 words -- no 19937-step Horner chain.
//  2. S workgroups per candidate (each owning a contiguous range of elements, reached by a doubling tree
//     over the sub-stream index) then walk both streams block by block, temper, convert word pairs to doubles
//     exactly as NumPy's legacy random_sample does ((a>>5)*2^26 + (b>>6)) / 2^53, and writes
//     H = (A - lambda*delta) + (psi*delta + ((u-.5)*psi)*0.15) with NumPy's rounding order.
#include "common.h"
#include "luws.h"
#include "mtjump.h"
#include <cstdint>
#include <cstdlib>

namespace {

constexpr int MTN = 624, MTM = 397, MTD = 227;          // MTD = MTN - MTM
constexpr uint32_t UPM = 0x80000000u, LOM = 0x7fffffffu, MAG = 0x9908b0dfu;
constexpr int PDEG = 19937;
constexpr int PWORDS = 624;                              // polynomial bit array, 32-bit words (19968 bits)
constexpr int CONVN = PDEG + MTN;                        // words of stream a jump needs (20561)
constexpr int CONV_BLOCKS = (CONVN + MTN - 1) / MTN;     // 33 blocks

__device__ __forceinline__ uint32_t twist(uint32_t u, uint32_t v) {
    const uint32_t y = (u & UPM) | (v & LOM);
    return (y >> 1) ^ ((y & 1u) ? MAG : 0u);
}

// Word k of the NEXT block from the CURRENT block `od` alone.  The reference recurrence (mt19937.cpp regen)
//   new[k] = new[k-227] ^ tw(od[k], od[k+1])      (k >= 227; new[k+397-624])
//   new[k] = od[k+397]  ^ tw(od[k], od[k+1])      (k <  227)
// is substituted into itself, so the words of the second and third part do not wait for the first:
//   227 <= k < 454:  new[k] = od[k+170] ^ tw(od[k-227], od[k-226]) ^ tw(od[k], od[k+1])
//   454 <= k < 624:  new[k] = od[k-57]  ^ tw(od[k-454], od[k-453]) ^ tw(od[k-227], od[k-226]) ^ tw(od[k], od[k+1])
// (for k = 623 the "od[k+1]" is new[0] = od[397] ^ tw(od[0], od[1])).  1-4 twists per word instead of one, and a
// whole block costs ONE barrier instead of three: used where a whole workgroup walks blocks with nothing else to do
// between barriers (the 33-block prologue of mt_jump_kernel).
__device__ __forceinline__ uint32_t next_word(const uint32_t* od, int k) {
    if (k < MTD) return od[k + MTM] ^ twist(od[k], od[k + 1]);
    if (k < 2 * MTD) return od[k + 170] ^ twist(od[k - MTD], od[k - MTD + 1]) ^ twist(od[k], od[k + 1]);
    const uint32_t nxt = (k == MTN - 1) ? (od[MTM] ^ twist(od[0], od[1])) : od[k + 1];
    return od[k - 57] ^ twist(od[k - 2 * MTD], od[k - 2 * MTD + 1]) ^ twist(od[k - MTD], od[k - MTD + 1]) ^ twist(od[k], nxt);
}

// next block `nw` from block `od` (both time-ordered in LDS, distinct buffers); all `nthreads` threads take part
__device__ __forceinline__ void regen_block(const uint32_t* od, uint32_t* nw, int tid, int nthreads) {
    for (int k = tid; k < MTN; k += nthreads) nw[k] = next_word(od, k);
    __syncthreads();
}

// states[g] <- g(F) states[g - src_off] for the generators listed in `sel` (src_off = 0: in place; the doubling tree over
// the sub-stream index reads the state of another generator, mtplan.cpp).  g is given as the list of its set
// coefficients (`taps`), so out[j] = XOR_t x[taps[t] + j] over the generator's own next 19937+624 words: a GF(2)
// convolution.  The tap indices are wave-uniform (scalar loads); each iteration issues 16 independent LDS reads.
// The kernel is bound by LDS bandwidth (~10 000 taps x 624 words per jump), so what matters is how many waves a CU has in
// flight: the stream is walked in TWO windows of 17 blocks (45 KB of LDS, three workgroups per CU) instead of one of 33
// (85 KB, one per CU; rounds 2-4).  Window 0 = blocks 0..16 serves the taps below JSPLIT (tap + 623 stays inside it),
// window 1 = blocks 16..32 the others, rebased by the host; either list is padded to a multiple of 16 with ZTAP, which
// points at a zero region.
constexpr int JT = 640;                                  // one output word per thread (624 used)
constexpr int JWIN = 17;                                 // blocks per window
constexpr int JSPLIT = (JWIN - 1) * MTN;                 // first tap of the second window
constexpr int XS_WORDS = JWIN * MTN;
constexpr int ZTAP = XS_WORDS;                           // xs[ZTAP .. ZTAP+JT) == 0
static_assert(JSPLIT + XS_WORDS >= CONVN, "two windows must cover the stream a jump needs");
__global__ void __launch_bounds__(JT)
mt_jump_kernel(uint32_t* states, const int* __restrict__ sel, const int* __restrict__ mult, MausJumpPolys P, int src_off)
{
    __shared__ uint32_t xs[XS_WORDS + JT];
    const int tid = threadIdx.x;
    const int v = mult ? mult[blockIdx.x] : 1;             // which polynomial of the launch: x^(v J)  (workgroup-uniform)
    const int* __restrict__ taps = P.taps[v];
    const int nlo16 = P.nlo16[v], ntap16 = P.ntap16[v];
    uint32_t* S = states + (long)sel[blockIdx.x] * MTN;
    const uint32_t* Src = S - (long)src_off * MTN;
    for (int k = tid; k < MTN; k += JT) xs[k] = Src[k];
    xs[ZTAP + tid] = 0u;
    __syncthreads();
    const uint32_t* xt = xs + tid;
    uint32_t acc0 = 0u, acc1 = 0u, acc2 = 0u, acc3 = 0u;
    int t = 0;
    for (int win = 0; win < 2; ++win) {
        for (int bk = 1; bk < JWIN; ++bk) regen_block(xs + (bk - 1) * MTN, xs + bk * MTN, tid, JT);
        const int t_end = win ? ntap16 : nlo16;
        for (; t < t_end; ++t) {
            const int4* tp = reinterpret_cast<const int4*>(taps + 16 * t);
            const int4 a = tp[0], b = tp[1], c = tp[2], d = tp[3];
            acc0 ^= xt[a.x] ^ xt[a.y] ^ xt[a.z] ^ xt[a.w];
            acc1 ^= xt[b.x] ^ xt[b.y] ^ xt[b.z] ^ xt[b.w];
            acc2 ^= xt[c.x] ^ xt[c.y] ^ xt[c.z] ^ xt[c.w];
            acc3 ^= xt[d.x] ^ xt[d.y] ^ xt[d.z] ^ xt[d.w];
        }
        if (win == 0) {                                     // block 16 becomes block 0 of the second window
            __syncthreads();
            const uint32_t carry = (tid < MTN) ? xs[JSPLIT + tid] : 0u;
            __syncthreads();
            if (tid < MTN) xs[tid] = carry;
            __syncthreads();
        }
    }
    if (tid < MTN) S[tid] = acc0 ^ acc1 ^ acc2 ^ acc3;
}

__global__ void mt_copy_state_kernel(uint32_t* __restrict__ states, const uint32_t* __restrict__ base, int count) {
    const int g = blockIdx.x;
    if (g < count) for (int k = threadIdx.x; k < MTN; k += blockDim.x) states[(long)g * MTN + k] = base[k];
}

__device__ __forceinline__ uint32_t temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// ---------------------------------------------------------------------------------------
// H build with regenerated draws, pipelined over the four waves of a workgroup (round 3).
// One workgroup per sub-stream: waves 0 / 1 PRODUCE the U1 / U2 stream, waves 2 / 3 CONSUME (temper, convert, build H).
//   * A producer owns one generator and keeps three blocks of it in LDS.  The next block is written out of place in 10
//     chunks of 64 words,  nw[k] = (k < 227 ? od[k+397] : nw[k-227]) ^ tw(od[k], od[k+1]):  a chunk only reads words of `nw`
//     that the same wave wrote at least three chunks earlier, and the LDS operations of one wave complete in order, so the
//     three dependency phases of a block need no barrier (round 2's kernel -- 320 threads per stream, three barriers per
//     624 words, every wave waiting for the slowest, a ring of doubles between production and consumption -- took 27 ms
//     per 181 matrices at n = 4096 where this one takes 15).
//   * In iteration t the producers write block t+2 while the consumers convert the 312 elements that block t is worth:
//     element q needs the words rpos + 2q, rpos + 2q + 1 counted from the start of block t, i.e. words of block t or
//     (past 623) of block t+1 -- both resident, so there is no ring buffer between producer and consumer, no lag
//     bookkeeping between the two streams (their positions in the block differ) and no pair straddling a block that
//     is gone.  One workgroup barrier per iteration.
//   * A 64-element chunk none of whose entries can be changed by the perturbation is copied: |p| <= 0.075 |psi| (|u - .5|
//     <= .5), and fl(a + p) = a whenever |p| < 2^(e-54) for a in [2^e, 2^(e+1)), so with thr = 0.075 |psi| 2^55 an entry
//     with |re|, |im| > thr off the diagonal is A's own bits.  At psi = 1e-20 and entries ~ N(0, 1/2n) that is 3 of 4
//     chunks at n = 4096; at the large psi of retries or stuck candidates every chunk takes the full path.  H is the same
//     bits either way (tests/test_gpu_mt19937.py compares with the host-drawn build at both kinds of psi).
// Against one wave per sub-stream doing everything (measured: 18.5 ms per 181 matrices at 32 sub-streams, but 44 ms of
// jump kernels to get 11 584 generator states) this needs a quarter of the generators for the same parallelism.
// grid = (G candidates, S sub-streams): blockIdx.x = candidate, so that the workgroups resident at any time read the same
// stretch of A (round 2's kernel fetched 0.7 x |A| from HBM per candidate, profiles/r02_pmc_traffic_per_kernel.txt).
// ---------------------------------------------------------------------------------------
constexpr int EPB = MTN / 2;                 // elements (doubles) per block and stream
constexpr int ECH = (EPB + 63) / 64;         // chunks of 64 elements per block (the last one holds 56)
constexpr int WCH = (MTN + 63) / 64;         // chunks of 64 words per block
constexpr int NBUF = 3;

// The 10 chunks fall into four groups -- chunks 0-2, 3-5, 6-8, 9 -- such that a chunk reads `nw` only where an EARLIER group
// wrote it (k - 227 lies at least 163 words back).  The compiler is told about the cross-lane dependence between groups
// only (wave_barrier: it reasons per thread, sees that nw[k-227] and nw[k'] never coincide for one lane and would move the
// load above the store); inside a group it is free to put all loads in flight together -- with a barrier behind every
// chunk a block cost ten LDS round trips in a row (1.7 us per iteration at four workgroups per candidate).
__device__ __forceinline__ void regen_wave(const uint32_t* od, uint32_t* nw, int lane) {
#pragma unroll
    for (int c = 0; c < WCH; ++c) {
        const int k = c * 64 + lane;
        if (k < MTN) {
            const uint32_t a = od[k];
            const uint32_t b = (k == MTN - 1) ? nw[0] : od[k + 1];
            const uint32_t m = (k < MTD) ? od[k + MTM] : nw[k - MTD];
            nw[k] = m ^ twist(a, b);
        }
        if (c % 3 == 2) __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_wave_barrier();
}
// chunk c reads nw[64c-227 .. 64c-164]; its group starts at word 192*(c/3) >= 64c - 128
static_assert(MTD - 63 > 2 * 64, "regen_wave: a chunk must read nw only below its own group of three chunks");

__global__ void __launch_bounds__(256)
build_h_mt4_kernel(const c128* __restrict__ A, int n, int npad, long ldh, long strideH, c128* __restrict__ Hg,
                   const c128* __restrict__ shift, const double* __restrict__ psi,
                   int rhs_mode, const c128* __restrict__ X, long ldx, const int* __restrict__ slots,
                   const c128* __restrict__ bvec,
                   const uint32_t* __restrict__ states /* [G][S][2][624] */, const int* __restrict__ extra /* [G][S][2] */,
                   const int* __restrict__ rpos /* [G][S][2] */, long E, int* __restrict__ flags, int tiled)
{
    __shared__ uint32_t blk[2][NBUF][MTN];   // [stream][buffer][word]
    const int g = blockIdx.x, sb = blockIdx.y, S = gridDim.y, tid = threadIdx.x, lane = tid & 63;
    // roles rotate with the workgroup: the producers of the workgroups that share a CU should not all sit on one SIMD
    const int role = ((tid >> 6) + g + sb) & 3;               // 0 / 1: producer of U1 / U2, 2 / 3: consumers
    const long gi = ((long)g * S + sb) * 2;                   // index of this workgroup's first generator
    c128* H = Hg + (long)g * strideH;
    auto hidx = [&](int i, int j) -> long { return tiled ? lu_tix(npad, i, j) : (long)i * ldh + j; };
    const c128 lam = shift[g];
    const double ps = psi[g];
    bool bad = false;

    // everything outside the n x n perturbation area (pad rows / columns, augmented block), shared by the S workgroups of the candidate
    {
        const long npadcols = (long)npad * (ldh - n);
        for (long e = (long)sb * 256 + tid; e < npadcols; e += (long)S * 256) {
            const int i = (int)(e / (ldh - n)), j = n + (int)(e - (long)i * (ldh - n));
            c128 v = cmake(0.0, 0.0);
            if (j < npad) { if (i == j) v.x = 1.0; }
            else if (j == npad && i < n) { v = (rhs_mode == 0) ? X[(long)slots[g] * ldx + i] : bvec[i]; bad |= !cfinite(v); }
            H[hidx(i, j)] = v;
        }
        for (long e = (long)sb * 256 + tid; e < (long)(npad - n) * n; e += (long)S * 256) {
            const int i = n + (int)(e / n), j = (int)(e - (long)(i - n) * n);
            H[hidx(i, j)] = cmake(0.0, 0.0);
        }
    }

    // each producer brings its stream to its start block and writes the block behind it
    const int ex0 = extra[gi], ex1 = extra[gi + 1];
    int c0 = ex0 % NBUF, c1 = ex1 % NBUF;                    // buffers holding block t of U1 / U2 (t = 0 now)
    if (role < 2) {
        uint32_t (*B)[MTN] = blk[role];
        const uint32_t* St = states + (gi + role) * MTN;
        for (int k = lane; k < MTN; k += 64) B[0][k] = St[k];
        __builtin_amdgcn_wave_barrier();
        const int ex = role ? ex1 : ex0;
        int cur = 0;
        for (int it = 0; it <= ex; ++it) { const int nx = (cur + 1 == NBUF) ? 0 : cur + 1; regen_wave(B[cur], B[nx], lane); cur = nx; }
    }
    __syncthreads();

    const int w0 = rpos[gi], w1 = rpos[gi + 1];      // first unread word of each stream's start block
    const long e0 = (long)sb * E;
    const long total = min((long)n * n, e0 + E);     // this workgroup's element range is [e0, total)
    const int niter = (int)((total - e0 + EPB - 1) / EPB);
    const int ci = role - 2;                         // consumer: chunks ci, ci + 2, (ci + 4) of every iteration
    constexpr int CPC = (ECH + 1) / 2;               // chunks per consumer and iteration, at most
    // row / column of the consumer's current chunk's first element (wave-uniform)
    int ib = 0, jb = 0;
    c128 apf[CPC];
    if (role >= 2) {
        const long ef = e0 + (long)ci * 64;
        ib = (int)(ef / n); jb = (int)(ef - (long)ib * n);
#pragma unroll
        for (int u = 0; u < CPC; ++u) apf[u] = A[max(0l, min(e0 + (ci + 2 * u) * 64 + lane, total - 1))];
    }
    // an entry with |re|, |im| above thr cannot be changed by the perturbation (see above); NaN / inf psi: nothing is skipped
    const double thr = fabs(ps) * (0.075 * 36028797018963968.0);
    for (int it = 0; it < niter; ++it) {
        const int n0 = (c0 + 1 == NBUF) ? 0 : c0 + 1, n1 = (c1 + 1 == NBUF) ? 0 : c1 + 1;
        if (role < 2) {
            // block it+2 from block it+1
            const int nx = role ? n1 : n0, nn = (nx + 1 == NBUF) ? 0 : nx + 1;
            regen_wave(blk[role][nx], blk[role][nn], lane);
        } else {
            const long e = e0 + (long)it * EPB;
            c128 anx[CPC];
#pragma unroll
            for (int u = 0; u < CPC; ++u) anx[u] = A[min(e + EPB + (ci + 2 * u) * 64 + lane, total - 1)];   // next iteration's entries
            const uint32_t* cu0 = blk[0][c0]; const uint32_t* nx0 = blk[0][n0];
            const uint32_t* cu1 = blk[1][c1]; const uint32_t* nx1 = blk[1][n1];
#pragma unroll
            for (int u = 0; u < CPC; ++u) {
                const int c = ci + 2 * u;
                if (c < ECH) {
                    const int q = c * 64 + lane;
                    int i = ib, j = jb + lane;
                    while (j >= n) { j -= n; ++i; }
                    const bool act = q < EPB && e + q < total;
                    const c128 a = apf[u];
                    const bool sens = !(fabs(a.x) > thr && fabs(a.y) > thr) || j == i;
                    c128 h = a;
                    if (__any(act && sens)) {
                        if (act) {
                            const int t0 = w0 + 2 * q, t1 = w1 + 2 * q;
                            const uint32_t a0 = (t0 < MTN) ? cu0[t0] : nx0[t0 - MTN], b0 = (t0 + 1 < MTN) ? cu0[t0 + 1] : nx0[t0 + 1 - MTN];
                            const uint32_t a1 = (t1 < MTN) ? cu1[t1] : nx1[t1 - MTN], b1 = (t1 + 1 < MTN) ? cu1[t1 + 1] : nx1[t1 + 1 - MTN];
                            const double u0 = ((double)(temper(a0) >> 5) * 67108864.0 + (double)(temper(b0) >> 6)) / 9007199254740992.0;
                            const double u1 = ((double)(temper(a1) >> 5) * 67108864.0 + (double)(temper(b1) >> 6)) / 9007199254740992.0;
                            const double pr = __dmul_rn(__dmul_rn(__dsub_rn(u0, 0.5), ps), 0.15);
                            const double pi = __dmul_rn(__dmul_rn(__dsub_rn(u1, 0.5), ps), 0.15);
                            if (j == i) {
                                h.x = __dadd_rn(__dsub_rn(a.x, lam.x), __dadd_rn(ps, pr));
                                h.y = __dadd_rn(__dsub_rn(a.y, lam.y), __dadd_rn(0.0, pi));
                            } else {
                                h.x = __dadd_rn(a.x, pr);
                                h.y = __dadd_rn(a.y, pi);
                            }
                        }
                    }
                    if (act) { bad |= !cfinite(h); H[hidx(i, j)] = h; }
                    // first element of this consumer's next chunk: two chunks on, or -- from its last chunk of the
                    // iteration -- chunk ci of the next block
                    const bool last = c + 2 >= ECH;
                    jb += last ? EPB - c * 64 + ci * 64 : 128;
                    while (jb >= n) { jb -= n; ++ib; }
                }
            }
#pragma unroll
            for (int u = 0; u < CPC; ++u) apf[u] = anx[u];
        }
        c0 = n0; c1 = n1;
        lds_barrier();            // LDS only: the consumers' loads of A and stores of H stay in flight across it
    }
    if (__any(bad) && lane == 0) atomicOr(&flags[g], 1);
}

}  // namespace

void maus_mt_copy_states(hipStream_t st, uint32_t* states, const uint32_t* base, int count) {
    hipLaunchKernelGGL(mt_copy_state_kernel, dim3(count), dim3(256), 0, st, states, base, count);
}
int maus_mt_zero_tap() { return ZTAP; }
int maus_mt_tap_split() { return JSPLIT; }
void maus_mt_jump(hipStream_t st, uint32_t* states, const int* sel, const int* mult, int nsel, const MausJumpPolys& P, int src_off) {
    if (nsel > 0) hipLaunchKernelGGL(mt_jump_kernel, dim3(nsel), dim3(JT), 0, st, states, sel, mult, P, src_off);
}
void maus_build_h_mt(hipStream_t st, const c128* A, int n, int npad, long ldh, long strideH, c128* H, int G, int S, long E,
                     const c128* d_shift, const double* d_psi, int rhs_mode, const c128* X, long ldx, const int* d_slots,
                     const c128* bvec, const uint32_t* states, const int* extra, const int* rpos, int* flags, int tiled) {
    hipLaunchKernelGGL(build_h_mt4_kernel, dim3(G, S), dim3(256), 0, st, A, n, npad, ldh, strideH, H, d_shift, d_psi, rhs_mode,
                       X, ldx, d_slots, bvec, states, extra, rpos, E, flags, tiled);
}
