// libmaus_hip C ABI (include/maus_hip.h): context, device memory, stream, and the batched
// phases of the MAUS candidate step.  Host orchestration stays in Python (ctypes).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ctx.h"
#include <unistd.h>

thread_local std::string g_err;

static void hist_free(maus_ctx* c);

int ensure_scalars(maus_ctx* c, int count) {
    if (count <= c->scal_cap) return 0;
    int cap = std::max(count, 2 * c->scal_cap);
    cap = std::max(cap, 64);
    void** ptrs[] = {(void**)&c->d_slots, (void**)&c->d_i1, (void**)&c->d_i2, (void**)&c->d_c1, (void**)&c->d_c2, (void**)&c->d_r1, (void**)&c->d_r2};
    for (auto p : ptrs) if (*p) { (void)hipFree(*p); *p = nullptr; }
    HIPCHK(c, hipMalloc((void**)&c->d_slots, sizeof(int) * cap));
    HIPCHK(c, hipMalloc((void**)&c->d_i1, sizeof(int) * cap));
    HIPCHK(c, hipMalloc((void**)&c->d_i2, sizeof(int) * cap));
    HIPCHK(c, hipMalloc((void**)&c->d_c1, sizeof(c128) * cap));
    HIPCHK(c, hipMalloc((void**)&c->d_c2, sizeof(c128) * cap));
    HIPCHK(c, hipMalloc((void**)&c->d_r1, sizeof(double) * cap * 4));
    HIPCHK(c, hipMalloc((void**)&c->d_r2, sizeof(double) * cap * 4));
    c->scal_cap = cap;
    return 0;
}

int ensure_scratch(maus_ctx* c, size_t bytes) {
    if (bytes <= c->scratch_bytes) return 0;
    // a quarter of headroom, and growth by at least a quarter: the population -- and with it the GMRES scratch -- grows by up to 15 candidates per iteration
    // (AMS:533-534), and an exact fit would free and map hundreds of MB in every loop body
    size_t want = std::max(bytes + bytes / 4, c->scratch_bytes + c->scratch_bytes / 4);
    if (c->scratch) { (void)hipFree(c->scratch); c->scratch = nullptr; c->scratch_bytes = 0; }
    if (hipMalloc(&c->scratch, want) != hipSuccess) { (void)hipGetLastError(); want = bytes; HIPCHK(c, hipMalloc(&c->scratch, want)); }
    c->scratch_bytes = want;
    return 0;
}

// ---- pinned staging --------------------------------------------------------------------------------------------------------
// A pageable buffer handed to hipMemcpy is pinned by the runtime and stays registered with the kernel driver.  When the caller
// later frees it (NumPy temporaries, the vectors of retired candidates, a matrix copy of the start-up diagnostics), the driver
// evicts the process's queues and re-validates every registered range before it restores them: 4-80 ms of idle GPU inside
// somebody's loop body, and launches that appear 30x slower than they are (tools/userptr_probe.py; the slow second loop body
// of BASELINE configs[2], the 20-40 ms first read-back of configs[3]).  A copy into fresh pageable memory is also staged page by
// page by the runtime (16 MB: 20 ms against 0.6 ms + a memcpy).  So the library owns one pinned buffer per context and every
// transfer of more than STAGE_DIRECT_MAX bytes goes through it; smaller ones are staged by the runtime itself.
static const size_t STAGE_DIRECT_MAX = 16u << 10;
// Small uploads are copied into a pinned ring and leave from there asynchronously; every entry point synchronises its streams
// before it returns and uses far less than the ring between two such points, so a slot is never rewritten while in flight.
static const size_t SMALL_RING = 2u << 20;

int maus_pin_ready(maus_ctx* c) {
    if (c->pin) return 0;
    if (c->pin_failed) return -1;
    const char* e = getenv("MAUS_PIN_BYTES");
    const size_t want = e ? std::max<size_t>(1u << 20, (size_t)atoll(e)) : ((size_t)32 << 20);
    if (hipHostMalloc(&c->pin, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); c->pin = nullptr; c->pin_failed = true; return -1; }
    c->pin_bytes = want;
    for (auto& ev : c->pin_ev) if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ev = nullptr; }
    if (!c->pin_ev[0] || !c->pin_ev[1] || hipHostMalloc((void**)&c->pin_small, SMALL_RING, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError(); (void)hipHostFree(c->pin); c->pin = nullptr; c->pin_small = nullptr; c->pin_bytes = 0; c->pin_failed = true; return -1;
    }
    return 0;
}

int maus_stage_h2d(maus_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return 0;
    if (maus_pin_ready(c)) {
        HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipStreamSynchronize(st));
        return 0;
    }
    const size_t half = c->pin_bytes / 2;
    size_t off = 0;
    for (int i = 0; off < bytes; ++i, off += half) {                // the memcpy of one half overlaps the DMA of the other
        const int b = i & 1;
        const size_t nb = std::min(half, bytes - off);
        if (i >= 2) HIPCHK(c, hipEventSynchronize(c->pin_ev[b]));
        memcpy((char*)c->pin + b * half, (const char*)src + off, nb);
        HIPCHK(c, hipMemcpyAsync((char*)dst + off, (char*)c->pin + b * half, nb, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipEventRecord(c->pin_ev[b], st));
    }
    HIPCHK(c, hipStreamSynchronize(st));
    return 0;
}

int maus_stage_d2h(maus_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return 0;
    if (maus_pin_ready(c)) {
        HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        return 0;
    }
    const size_t half = c->pin_bytes / 2;
    size_t off = 0, prev_off = 0, prev_nb = 0;
    int i = 0;
    for (; off < bytes; ++i, off += half) {
        const int b = i & 1;
        const size_t nb = std::min(half, bytes - off);
        HIPCHK(c, hipMemcpyAsync((char*)c->pin + b * half, (const char*)src + off, nb, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipEventRecord(c->pin_ev[b], st));
        if (i >= 1) {                                               // drain the other half while this one is in flight
            HIPCHK(c, hipEventSynchronize(c->pin_ev[b ^ 1]));
            memcpy((char*)dst + prev_off, (char*)c->pin + (b ^ 1) * half, prev_nb);
        }
        prev_off = off; prev_nb = nb;
    }
    const int lb = (i - 1) & 1;
    HIPCHK(c, hipEventSynchronize(c->pin_ev[lb]));
    memcpy((char*)dst + prev_off, (char*)c->pin + lb * half, prev_nb);
    return 0;
}

// Small transfers stay asynchronous (the runtime stages them itself; the caller synchronises as before), larger ones are staged
// and complete on return.
int maus_h2d(maus_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return 0;
    if (bytes > STAGE_DIRECT_MAX) return maus_stage_h2d(c, dst, src, bytes, st);
    if (maus_pin_ready(c)) { HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st)); return 0; }
    // The ring is reused in two halves, each guarded by an event: before a half is written again, the event recorded behind the
    // last upload of its previous lap is waited for.  Every entry point synchronises its streams long before 2 MiB of small
    // uploads are out, so these waits never block on today's paths -- but nothing else enforced that.
    const size_t need = (bytes + 63) & ~(size_t)63;
    for (auto& e : c->pin_small_ev) if (!e) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (c->pin_small_off + need > SMALL_RING) {                      // wrap: the second half is complete, the first is entered
        HIPCHK(c, hipEventRecord(c->pin_small_ev[1], st));
        if (c->pin_small_lap > 0 || c->pin_small_mid) HIPCHK(c, hipEventSynchronize(c->pin_small_ev[0]));
        c->pin_small_off = 0; c->pin_small_lap++; c->pin_small_mid = false;
    }
    const size_t before = c->pin_small_off;
    const bool crossing = before < SMALL_RING / 2 && before + need >= SMALL_RING / 2;
    if (crossing && c->pin_small_lap > 0) HIPCHK(c, hipEventSynchronize(c->pin_small_ev[1]));     // the second half is entered
    char* slot = c->pin_small + before;
    c->pin_small_off = before + need;
    memcpy(slot, src, bytes);
    HIPCHK(c, hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, st));
    if (crossing) { HIPCHK(c, hipEventRecord(c->pin_small_ev[0], st)); c->pin_small_mid = true; }    // the first half is complete
    return 0;
}
// complete on return (the destination is filled by a memcpy from the pinned buffer)
int maus_d2h(maus_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return 0;
    if (maus_pin_ready(c)) { HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st)); HIPCHK(c, hipStreamSynchronize(st)); return 0; }
    if (bytes > c->pin_bytes / 2) return maus_stage_d2h(c, dst, src, bytes, st);
    HIPCHK(c, hipMemcpyAsync(c->pin, src, bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    memcpy(dst, c->pin, bytes);
    return 0;
}

int check_slots(maus_ctx* c, const int* slots, int count) {
    if (count < 0) FAIL(c, "negative count");
    for (int i = 0; i < count; ++i) if (slots[i] < 0 || slots[i] >= c->cap) FAIL(c, "slot out of range");
    return 0;
}

int upload_slots(maus_ctx* c, const int* slots, int count) {
    if (check_slots(c, slots, count)) return -1;
    if (ensure_scalars(c, count)) return -1;
    if (maus_h2d(c, c->d_slots, slots, sizeof(int) * count, c->st)) return -1;
    return 0;
}

// ---- which rows of Y still hold A X (see ctx.h) ----------------------------------------------------------------------------
// Every entry point that writes X or Y, changes A or re-allocates the population drops all stamps (a new epoch); maus_pop_put
// on X drops the rows it writes; maus_residual (SVD) stamps the rows it has just computed; maus_svd_power_propose skips the
// product for stamped rows.  Read-only entry points (pop_get, hist_*, gram, profile) leave the stamps alone.
static inline void av_drop_all(maus_ctx* c) { maus_av_drop_all(c); }
static inline void av_drop(maus_ctx* c, const int* slots, int count) {
    for (int k = 0; k < count; ++k) if (slots[k] >= 0 && (size_t)slots[k] < c->av_stamp.size()) c->av_stamp[slots[k]] = 0;
}
// A stamp promises more than "this row holds A x": it promises the bits that a product over MORE than 32 rows on the DMA 3M
// kernels gives -- what a later, larger call would compute for the row.  Products of up to 32 rows (and of matrices the DMA
// staging does not take: K not a multiple of 8) run other kernels with another rounding, so they leave no stamps, and reuse is
// only attempted by calls that are themselves in that family.
static inline bool av_family(const maus_ctx* c, int count, int K) { return count > 32 && (K % 8) == 0 && K >= 64; }
static inline void av_mark(maus_ctx* c, const int* slots, int count) {
    if (!av_family(c, count, c->cols)) return;
    if (c->av_stamp.size() < (size_t)c->cap) c->av_stamp.resize(c->cap, 0u);
    for (int k = 0; k < count; ++k) c->av_stamp[slots[k]] = c->av_epoch;
}
static inline bool av_has(const maus_ctx* c, int slot) { return (size_t)slot < c->av_stamp.size() && c->av_stamp[slot] == c->av_epoch; }
// the same for S = A^H U (ctx.h): stamped by maus_svd_commit for the rows whose u it has just installed, dropped with Y's stamps
// and by maus_pop_put on U
static inline void ahu_drop(maus_ctx* c, const int* slots, int count) {
    for (int k = 0; k < count; ++k) if (slots[k] >= 0 && (size_t)slots[k] < c->ahu_stamp.size()) c->ahu_stamp[slots[k]] = 0;
}
static inline void ahu_mark(maus_ctx* c, const int* slots, int count) {
    if (c->ahu_stamp.size() < (size_t)c->cap) c->ahu_stamp.resize(c->cap, 0u);
    for (int k = 0; k < count; ++k) c->ahu_stamp[slots[k]] = c->ahu_epoch;
}
static inline bool ahu_has(const maus_ctx* c, int slot) { return c->S && (size_t)slot < c->ahu_stamp.size() && c->ahu_stamp[slot] == c->ahu_epoch; }

// profiling hook handed to the LU driver
void prof_tick(void* ud, int klass, int phase, double flops, double bytes) {
    maus_ctx* c = (maus_ctx*)ud;
    if (!c->prof_on) return;
    if (phase == 0) {
        c->total_launches[klass]++;
        // Sampled mode: every prof_stride_big-th K>=256 zgemm launch (the trailing updates proper) is bracketed and
        // stands for `stride` launches (weight).  The short launches are left alone: with two sub-batch streams in
        // flight their event-to-event time is mostly the other stream's kernels (prof_stride_small > 0 samples
        // them anyway, for single-stream runs).
        const bool is_gemm = (klass == KC_GEMM || klass >= KC_GEMM_K128);
        const int stride = (klass == KC_GEMM) ? c->prof_stride_big : c->prof_stride_small;
        c->prof_weight = (c->prof_mode == 2) ? (double)stride : 1.0;
        c->prof_skip = (c->prof_mode == 2) && (!is_gemm || stride <= 0 || (c->prof_cnt[klass]++ % stride) != 0);
    }
    if (c->prof_skip) return;
    auto get = [&]() { hipEvent_t e; if (!c->pool.empty()) { e = c->pool.back(); c->pool.pop_back(); } else { (void)hipEventCreate(&e); } return e; };
    hipStream_t st = c->prof_st ? c->prof_st : c->st;
    if (phase == 0) { c->cur0 = get(); (void)hipEventRecord(c->cur0, st); }
    else { hipEvent_t e1 = get(); (void)hipEventRecord(e1, st); c->pending.push_back({klass, c->cur0, e1, flops, bytes, c->prof_weight}); c->cur0 = nullptr; }
}

static void prof_resolve(maus_ctx* c) {
    if (c->pending.empty()) return;
    (void)hipStreamSynchronize(c->st);
    for (auto& r : c->pending) {
        float ms = 0.f; (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        if (c->prof_origin) {                            // interval on the common clock, for the union over both streams
            float t0 = 0.f;
            if (hipEventElapsedTime(&t0, c->prof_origin, r.e0) == hipSuccess) c->prof_iv[r.klass].push_back({t0, t0 + ms});
        }
        c->launches[r.klass] += (long)r.weight; c->ms[r.klass] += ms * r.weight; c->flops[r.klass] += r.flops * r.weight;
        c->bytes[r.klass] += r.bytes * r.weight;
        c->pool.push_back(r.e0); c->pool.push_back(r.e1);
    }
    c->pending.clear();
}


extern "C" {

int maus_abi_version(void) { return 1; }

int maus_ctx_create(int device, maus_ctx** out) {
    if (!out) return -1;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { g_err = "no such HIP device"; return -1; }
    if (hipSetDevice(device) != hipSuccess) { g_err = "hipSetDevice failed"; return -1; }
    maus_ctx* c = new maus_ctx();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) != hipSuccess) { delete c; g_err = "hipStreamCreate failed"; return -1; }
    (void)hipEventCreate(&c->t0); (void)hipEventCreate(&c->t1);
    *out = c;
    return 0;
}

int maus_ctx_destroy(maus_ctx* c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->st);
    (void)maus_comm_destroy(c);
    void* ptrs[] = {c->A, c->b, c->V, c->X, c->U, c->W, c->Y, c->d_slots, c->d_i1, c->d_i2, c->d_c1, c->d_c2, c->d_r1, c->d_r2,
                    c->H, c->ipiv, c->perm, c->ident, c->mw_sync, c->info, c->flags, c->Upert, c->scratch, c->hq, c->htau, c->hz, c->S};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->pin_small) (void)hipHostFree(c->pin_small);
    for (auto e : c->pin_small_ev) if (e) (void)hipEventDestroy(e);
    for (auto ev : c->pin_ev) if (ev) (void)hipEventDestroy(ev);
    hist_free(c);
    for (auto& kv : c->mt_taps) if (kv.second.taps) (void)hipFree(kv.second.taps);
    for (auto& b : c->mt_bufs) { if (b.states) (void)hipFree(b.states); if (b.ints) (void)hipFree(b.ints); if (b.base) (void)hipFree(b.base); }
    for (auto& r : c->pending) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (auto e : c->pool) (void)hipEventDestroy(e);
    for (auto st : c->lu_st) (void)hipStreamDestroy(st);
    for (auto e : c->lu_done) (void)hipEventDestroy(e);
    if (c->ev_stage) (void)hipEventDestroy(c->ev_stage);
    if (c->t0) (void)hipEventDestroy(c->t0);
    if (c->t1) (void)hipEventDestroy(c->t1);
    if (c->prof_origin) (void)hipEventDestroy(c->prof_origin);
    (void)hipStreamDestroy(c->st);
    delete c;
    return 0;
}

const char* maus_last_error(const maus_ctx* c) { return c ? c->err.c_str() : g_err.c_str(); }

int maus_device_info(maus_ctx* c, char* name, int name_len, int* cus, size_t* hbm_total, size_t* hbm_free) {
    hipDeviceProp_t p;
    HIPCHK(c, hipGetDeviceProperties(&p, c->device));
    if (name && name_len > 0) { snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName); }
    if (cus) *cus = p.multiProcessorCount;
    size_t fr = 0, tot = 0;
    HIPCHK(c, hipMemGetInfo(&fr, &tot));
    if (hbm_total) *hbm_total = tot;
    if (hbm_free) *hbm_free = fr;
    return 0;
}

int maus_sync(maus_ctx* c) { HIPCHK(c, hipStreamSynchronize(c->st)); return 0; }

static void free_population(maus_ctx* c) {
    c128** ps[] = {&c->X, &c->U, &c->W, &c->Y};
    for (auto p : ps) if (*p) { (void)hipFree(*p); *p = nullptr; }
    c->cap = 0; c->ldp = 0;
}

}   // extern "C"

// room for a rows x cols problem matrix; everything that belonged to the previous one goes (also called by comm.hip for the
// ranks that receive the matrix device to device)
int maus_matrix_reserve(maus_ctx* c, int rows, int cols) {
    av_drop_all(c);
    if (rows <= 0 || cols <= 0) FAIL(c, "maus_set_matrix: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->st));
    if (rows != c->rows || cols != c->cols) {
        if (c->A) { (void)hipFree(c->A); c->A = nullptr; }
        if (std::max(rows, cols) != c->ldp) { free_population(c); hist_free(c); }      // vector length changed
        if (c->V) { (void)hipFree(c->V); c->V = nullptr; c->vn = 0; }
        HIPCHK(c, hipMalloc((void**)&c->A, sizeof(c128) * (size_t)rows * cols));
        c->rows = rows; c->cols = cols;
    }
    if (c->hq) { (void)hipFree(c->hq); c->hq = nullptr; } if (c->htau) { (void)hipFree(c->htau); c->htau = nullptr; } c->hqn = 0;   // reflectors of the previous matrix
    if (c->hz) { (void)hipFree(c->hz); c->hz = nullptr; c->hzn = 0; }
    return 0;
}

extern "C" {

int maus_set_matrix(maus_ctx* c, const double* a, int rows, int cols) {
    if (!a) FAIL(c, "maus_set_matrix: bad arguments");
    if (maus_matrix_reserve(c, rows, cols)) return -1;
    return maus_stage_h2d(c, c->A, a, sizeof(c128) * (size_t)rows * cols, c->st);
}

int maus_set_rhs(maus_ctx* c, const double* b, int n) {
    if (!b || n <= 0) FAIL(c, "maus_set_rhs: bad arguments");
    HIPCHK(c, hipStreamSynchronize(c->st));
    if (n != c->bn) { if (c->b) (void)hipFree(c->b); c->b = nullptr; HIPCHK(c, hipMalloc((void**)&c->b, sizeof(c128) * n)); c->bn = n; }
    return maus_stage_h2d(c, c->b, b, sizeof(c128) * n, c->st);
}

int maus_set_eigvecs(maus_ctx* c, const double* v, int n) {
    if (!v || n <= 0 || n != c->rows || n != c->cols) FAIL(c, "maus_set_eigvecs: n must match the square problem matrix");
    HIPCHK(c, hipStreamSynchronize(c->st));
    if (n != c->vn) { if (c->V) (void)hipFree(c->V); c->V = nullptr; HIPCHK(c, hipMalloc((void**)&c->V, sizeof(c128) * (size_t)n * n)); c->vn = n; }
    return maus_stage_h2d(c, c->V, v, sizeof(c128) * (size_t)n * n, c->st);
}

int maus_get_eigvecs(maus_ctx* c, double* v_out, int n) {
    if (!c->V || c->vn != n || !v_out) FAIL(c, "maus_get_eigvecs: no eigenvector matrix of that order on the device");
    HIPCHK(c, hipStreamSynchronize(c->st));
    return maus_stage_d2h(c, v_out, c->V, sizeof(c128) * (size_t)n * n, c->st);
}

int maus_pop_capacity(maus_ctx* c) { return c->cap; }

int maus_pop_reserve(maus_ctx* c, int capacity) {
    av_drop_all(c);
    if (c->rows <= 0) FAIL(c, "maus_pop_reserve: set the matrix first");
    if (capacity <= c->cap) return 0;
    HIPCHK(c, hipStreamSynchronize(c->st));
    long ld = std::max(c->rows, c->cols);
    int newcap = std::max(capacity, c->cap + c->cap / 2);
    c128** ps[] = {&c->X, &c->U, &c->W, &c->Y};
    for (auto p : ps) {
        c128* nw = nullptr;
        HIPCHK(c, hipMalloc((void**)&nw, sizeof(c128) * (size_t)newcap * ld));
        HIPCHK(c, hipMemset(nw, 0, sizeof(c128) * (size_t)newcap * ld));
        if (*p) { HIPCHK(c, hipMemcpy(nw, *p, sizeof(c128) * (size_t)c->cap * ld, hipMemcpyDeviceToDevice)); (void)hipFree(*p); }
        *p = nw;
    }
    c->cap = newcap; c->ldp = ld;
    // First-use costs of the read-back paths belong here, not in the loop body in which the first candidates converge: the
    // runtime loads its strided-copy kernel at the first hipMemcpy2D (8 ms measured), and the gather scratch of maus_pop_get /
    // maus_gram would otherwise be (re)allocated there.
    if (ensure_scratch(c, sizeof(c128) * (size_t)std::min(newcap, 256) * ld)) return -1;
    (void)maus_pin_ready(c);                                        // without it the direct copies remain
    { std::vector<c128> tmp(2 * 8);
      HIPCHK(c, hipMemcpy2DAsync(tmp.data(), sizeof(c128) * 8, c->X, sizeof(c128) * ld, sizeof(c128) * std::min<long>(8, ld), std::min(2, newcap), hipMemcpyDeviceToHost, c->st));
      HIPCHK(c, hipStreamSynchronize(c->st)); }
    return 0;
}

static c128* pop_array(maus_ctx* c, int which) {
    switch (which) { case MAUS_POP_X: return c->X; case MAUS_POP_U: return c->U; case MAUS_POP_W: return c->W; case MAUS_POP_Y: return c->Y; }
    return nullptr;
}

static bool contiguous(const int* slots, int count) {
    for (int i = 1; i < count; ++i) if (slots[i] != slots[0] + i) return false;
    return true;
}

// Candidate vectors cross the bus through the context's pinned buffer (maus_pop_reserve), never straight from / into the
// caller's arrays.  Two reasons, both measured (tools/popget_probe.py, tools/userptr_probe.py): a copy into fresh pageable
// memory is staged page by page by the runtime (16 MB: 20 ms against 0.6 ms + a memcpy), and a caller buffer handed to
// hipMemcpy is registered with the driver -- when NumPy later frees it, the next submission of this process waits for the
// driver's invalidation work (4-40 ms of idle GPU inside somebody's loop body).
int maus_pop_put(maus_ctx* c, int which, const int* slots, int count, const double* host, int len) {
    if (which == MAUS_POP_X) av_drop(c, slots, count); else if (which == MAUS_POP_U) ahu_drop(c, slots, count); else if (which == MAUS_POP_Y) av_drop_all(c);
    c128* P = pop_array(c, which);
    if (!P) FAIL(c, "maus_pop_put: population not reserved / bad array id");
    if (len <= 0 || len > c->ldp) FAIL(c, "maus_pop_put: bad vector length");
    if (check_slots(c, slots, count)) return -1;
    if (count == 0) return 0;
    const size_t rowb = sizeof(c128) * (size_t)len;
    if (!c->pin || c->pin_bytes < rowb) {                          // no staging buffer: straight from the caller's memory
        if (contiguous(slots, count)) {
            HIPCHK(c, hipMemcpy2DAsync(P + (long)slots[0] * c->ldp, sizeof(c128) * c->ldp, host, rowb, rowb, count, hipMemcpyHostToDevice, c->st));
        } else {
            for (int i = 0; i < count; ++i)
                HIPCHK(c, hipMemcpyAsync(P + (long)slots[i] * c->ldp, host + 2 * (size_t)i * len, rowb, hipMemcpyHostToDevice, c->st));
        }
        HIPCHK(c, hipStreamSynchronize(c->st));
        return 0;
    }
    const int per = (int)(c->pin_bytes / rowb);
    for (int off = 0; off < count; off += per) {
        const int k = std::min(per, count - off);
        memcpy(c->pin, host + 2 * (size_t)off * len, rowb * k);
        if (contiguous(slots + off, k)) {
            if (k == 1) HIPCHK(c, hipMemcpyAsync(P + (long)slots[off] * c->ldp, c->pin, rowb, hipMemcpyHostToDevice, c->st));
            else HIPCHK(c, hipMemcpy2DAsync(P + (long)slots[off] * c->ldp, sizeof(c128) * c->ldp, c->pin, rowb, rowb, k, hipMemcpyHostToDevice, c->st));
        } else {
            for (int i = 0; i < k; ++i)
                HIPCHK(c, hipMemcpyAsync(P + (long)slots[off + i] * c->ldp, (const char*)c->pin + rowb * i, rowb, hipMemcpyHostToDevice, c->st));
        }
        HIPCHK(c, hipStreamSynchronize(c->st));                     // the buffer is reused by the next chunk / call
    }
    return 0;
}

// rows `slots` of a population array, packed (maus_pop_get, maus_gram)
__global__ void gather_rows_kernel(const c128* __restrict__ X, long ldx, const int* __restrict__ slots, int len,
                                   c128* __restrict__ out) {
    const c128* src = X + (long)slots[blockIdx.x] * ldx;
    c128* dst = out + (long)blockIdx.x * len;
    for (int k = threadIdx.x; k < len; k += blockDim.x) dst[k] = src[k];
}

int maus_pop_get(maus_ctx* c, int which, const int* slots, int count, double* host, int len) {
    c128* P = pop_array(c, which);
    if (!P) FAIL(c, "maus_pop_get: population not reserved / bad array id");
    if (len <= 0 || len > c->ldp) FAIL(c, "maus_pop_get: bad vector length");
    if (check_slots(c, slots, count)) return -1;
    if (count == 0) return 0;
    const size_t rowb = sizeof(c128) * (size_t)len;
    if (!c->pin || c->pin_bytes < rowb + sizeof(int) * 4) {        // no staging buffer: straight into the caller's memory
        if (contiguous(slots, count)) {
            HIPCHK(c, hipMemcpy2DAsync(host, rowb, P + (long)slots[0] * c->ldp, sizeof(c128) * c->ldp, rowb, count, hipMemcpyDeviceToHost, c->st));
        } else {
            for (int i = 0; i < count; ++i)
                HIPCHK(c, hipMemcpyAsync(host + 2 * (size_t)i * len, P + (long)slots[i] * c->ldp, rowb, hipMemcpyDeviceToHost, c->st));
        }
        HIPCHK(c, hipStreamSynchronize(c->st));
        return 0;
    }
    const int per = (int)(c->pin_bytes / rowb);
    for (int off = 0; off < count; off += per) {
        const int k = std::min(per, count - off);
        if (k == 1) {
            HIPCHK(c, hipMemcpyAsync(c->pin, P + (long)slots[off] * c->ldp, rowb, hipMemcpyDeviceToHost, c->st));
        } else if (contiguous(slots + off, k)) {
            HIPCHK(c, hipMemcpy2DAsync(c->pin, rowb, P + (long)slots[off] * c->ldp, sizeof(c128) * c->ldp, rowb, k, hipMemcpyDeviceToHost, c->st));
        } else if (k < 4) {
            for (int i = 0; i < k; ++i)
                HIPCHK(c, hipMemcpyAsync((char*)c->pin + rowb * i, P + (long)slots[off + i] * c->ldp, rowb, hipMemcpyDeviceToHost, c->st));
        } else {                                                    // scattered rows: gather on the device, one copy
            if (ensure_scalars(c, k)) return -1;
            if (ensure_scratch(c, rowb * k)) return -1;
            memcpy(c->pin, slots + off, sizeof(int) * k);            // (consumed by the gather before the rows land on it)
            HIPCHK(c, hipMemcpyAsync(c->d_slots, c->pin, sizeof(int) * k, hipMemcpyHostToDevice, c->st));
            hipLaunchKernelGGL(gather_rows_kernel, dim3(k), dim3(256), 0, c->st, P, c->ldp, c->d_slots, len, (c128*)c->scratch);
            HIPCHK(c, hipMemcpyAsync(c->pin, c->scratch, rowb * k, hipMemcpyDeviceToHost, c->st));
        }
        HIPCHK(c, hipStreamSynchronize(c->st));
        memcpy(host + 2 * (size_t)off * len, c->pin, rowb * k);
    }
    return 0;
}

__global__ void copy_rows_kernel(c128* __restrict__ dst, const c128* __restrict__ src, long ld, const int* __restrict__ slots, int len) {
    const long o = (long)slots[blockIdx.x] * ld;
    for (int k = threadIdx.x; k < len; k += blockDim.x) dst[o + k] = src[o + k];
}

int maus_pop_device_ptr(maus_ctx* c, int which, void** ptr_out, long* ld_out, int* capacity_out) {
    av_drop_all(c);
    c128* P = pop_array(c, which);
    if (!P || !ptr_out) FAIL(c, "maus_pop_device_ptr: population not reserved / bad arguments");
    HIPCHK(c, hipStreamSynchronize(c->st));          // the caller is about to touch the rows from another stream
    *ptr_out = (void*)P;
    if (ld_out) *ld_out = c->ldp;
    if (capacity_out) *capacity_out = c->cap;
    return 0;
}

int maus_pop_copy(maus_ctx* c, int which_dst, int which_src, const int* slots, int count) {
    av_drop_all(c);
    c128* D = pop_array(c, which_dst); c128* S = pop_array(c, which_src);
    if (!D || !S || D == S) FAIL(c, "maus_pop_copy: population not reserved / bad array ids");
    if (count == 0) return 0;
    if (upload_slots(c, slots, count)) return -1;
    hipLaunchKernelGGL(copy_rows_kernel, dim3(count), dim3(256), 0, c->st, D, S, c->ldp, c->d_slots, (int)c->ldp);
    HIPCHK(c, hipStreamSynchronize(c->st));
    return 0;
}

// ---- history store (AMS:126, 303-304: param_history keeps every iterate of every candidate) -------------------------
// Rows are appended device-to-device (no PCIe traffic in the step) into fixed chunks that are never moved; a row is
// addressed by its append index.  Beyond MAUS_HIST_DEVICE_BYTES (default 1/8 of the device memory, at least 8 GiB) the oldest device chunk is spilled to
// host memory, so the history can grow like the reference's Python lists do without taking HBM from the LU workspace.
__global__ void hist_append_kernel(c128* __restrict__ dst, long len, const c128* __restrict__ src, long ld, const int* __restrict__ slots) {
    const c128* s = src + (long)slots[blockIdx.x] * ld;
    c128* d = dst + (long)blockIdx.x * len;
    for (long k = threadIdx.x; k < len; k += blockDim.x) d[k] = s[k];
}

static void hist_free(maus_ctx* c) {
    for (auto& h : c->hist) { if (h.dev) (void)hipFree(h.dev); if (h.host) free(h.host); }
    if (!c->hist.empty() || c->hist_rows) c->hist_gen++;
    c->hist.clear(); c->hist_len = 0; c->hist_rows = 0; c->hist_dev_bytes = 0;
}

int maus_hist_clear(maus_ctx* c) { HIPCHK(c, hipStreamSynchronize(c->st)); hist_free(c); return 0; }

int64_t maus_hist_generation(maus_ctx* c) { return c ? (int64_t)c->hist_gen : -1; }

int maus_hist_append(maus_ctx* c, int which, const int* slots, int count, int len, int64_t* first_index_out) {
    c128* P = pop_array(c, which);
    if (!P) FAIL(c, "maus_hist_append: population not reserved / bad array id");
    if (len <= 0 || len > c->ldp) FAIL(c, "maus_hist_append: bad vector length");
    if (check_slots(c, slots, count)) return -1;
    if (first_index_out) *first_index_out = c->hist_rows;
    if (count == 0) return 0;
    len = (int)c->ldp;                                               // rows are stored whole (u and v of an SVD problem differ in length)
    c->hist_len = len;
    // device budget of the store: MAUS_HIST_DEVICE_BYTES, default 1/8 of the device's memory but at least 8 GiB (36 GB on an
    // MI355X: BASELINE configs[4] appends 400 MB per loop body, and spilling a chunk to pageable host memory costs 20-40 ms)
    static const size_t budget_default = [] { size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); tot = 0; }
                                              return std::max((size_t)8 << 30, tot / 8); }();
    const char* be = getenv("MAUS_HIST_DEVICE_BYTES");
    const size_t budget = be ? (size_t)atoll(be) : budget_default;
    const char* ce = getenv("MAUS_HIST_CHUNK_BYTES");                                                          // 256 MiB chunks
    const long chunk_rows = !c->hist.empty() ? c->hist.front().cap
                          : std::max<long>(4, (long)((ce ? (size_t)atoll(ce) : ((size_t)256 << 20)) / (sizeof(c128) * (size_t)len)));
    int done = 0;
    while (done < count) {
        if (c->hist.empty() || c->hist.back().rows == c->hist.back().cap) {
            maus_ctx::HistChunk h; h.cap = chunk_rows;
            const size_t bytes = sizeof(c128) * (size_t)len * h.cap;
            while (c->hist_dev_bytes + bytes > budget) {             // spill the oldest device chunk to the host
                maus_ctx::HistChunk* old = nullptr;
                for (auto& q : c->hist) if (q.dev) { old = &q; break; }
                if (!old) break;
                const size_t ob = sizeof(c128) * (size_t)len * old->cap;
                old->host = (c128*)malloc(ob);
                if (!old->host) FAIL(c, "maus_hist_append: out of host memory");
                HIPCHK(c, hipStreamSynchronize(c->st));
                if (maus_stage_d2h(c, old->host, old->dev, sizeof(c128) * (size_t)len * old->rows, c->st)) return -1;
                (void)hipFree(old->dev); old->dev = nullptr; c->hist_dev_bytes -= ob;
            }
            HIPCHK(c, hipMalloc((void**)&h.dev, bytes));
            c->hist_dev_bytes += bytes;
            c->hist.push_back(h);
        }
        auto& h = c->hist.back();
        const int take = (int)std::min<long>(count - done, h.cap - h.rows);
        if (upload_slots(c, slots + done, take)) return -1;
        hipLaunchKernelGGL(hist_append_kernel, dim3(take), dim3(256), 0, c->st, h.dev + (size_t)h.rows * len, (long)len, P, c->ldp, c->d_slots);
        HIPCHK(c, hipStreamSynchronize(c->st));                       // d_slots is reused by the next call
        h.rows += take; done += take; c->hist_rows += take;
    }
    return 0;
}

int maus_hist_get(maus_ctx* c, const int64_t* indices, int count, int len, double* host_c128) {
    if (count < 0 || len <= 0 || len > c->hist_len) FAIL(c, "maus_hist_get: bad arguments");
    const long chunk_rows = c->hist.empty() ? 1 : c->hist.front().cap;
    for (int i = 0; i < count; ++i) {
        const int64_t ix = indices[i];
        if (ix < 0 || ix >= c->hist_rows) FAIL(c, "maus_hist_get: index out of range");
        const auto& h = c->hist[(size_t)(ix / chunk_rows)];
        const size_t off = (size_t)(ix % chunk_rows) * c->hist_len;
        double* out = host_c128 + 2 * (size_t)i * len;
        if (h.dev) { if (maus_stage_d2h(c, out, h.dev + off, sizeof(c128) * len, c->st)) return -1; }
        else memcpy(out, h.host + off, sizeof(c128) * len);
    }
    return 0;
}

// Y[slot] = A @ X[slot] for all listed slots:  C[count, rows] = Xg[count, cols] * A^T  (A as [n][k])
static void matvec_into_Y(maus_ctx* c, const c128* src, int count) {
    ProfScope ps(c, KC_GEMM, 8.0 * count * c->rows * c->cols, 16.0 * ((double)c->rows * c->cols + 2.0 * count * c->ldp));
    maus_zgemm_launch_idx(c->st, count, c->rows, c->cols, src, c->ldp, 0, c->A, c->cols, 0, c->Y, c->ldp, 0,
                          1.0, 0, 1, /*blay*/1, false, false, c->d_slots, c->d_slots);
}

// Y = A X for the rows of `slots` that do not hold it already (stamps: av_*).  The rows whose product an earlier call left in Y
// (same x, same product, same kernel family: the same bits) are not multiplied again; a handful of missing rows is topped up to
// 33 so that the partial product takes the DMA 3M kernels like the full one (per-element arithmetic does not depend on the tile
// shape: tests/test_gpu_kernels.py).  Leaves d_slots = slots.
static int matvec_missing_into_Y(maus_ctx* c, const int* slots, int count) {
    std::vector<int> need;
    for (int k = 0; k < count; ++k) if (!av_has(c, slots[k])) need.push_back(slots[k]);
    if ((int)need.size() == count || !av_family(c, count, c->cols)) { matvec_into_Y(c, c->X, count); return 0; }
    if (need.empty()) return 0;
    for (int k = 0; k < count && need.size() < 33; ++k) if (av_has(c, slots[k])) need.push_back(slots[k]);
    if (need.size() < 33) { matvec_into_Y(c, c->X, count); return 0; }      // (cannot happen with count > 32; kept for the invariant)
    if (upload_slots(c, need.data(), (int)need.size())) return -1;
    matvec_into_Y(c, c->X, (int)need.size());
    return upload_slots(c, slots, count);
}

int maus_matvec_rayleigh(maus_ctx* c, const int* slots, int count, double* num, double* den) {
    if (!c->A || !c->X) FAIL(c, "maus_matvec_rayleigh: matrix/population missing");
    if (c->rows != c->cols) FAIL(c, "maus_matvec_rayleigh: square matrix required");
    if (count == 0) return 0;
    if (upload_slots(c, slots, count)) return -1;
    // (the residual of the previous loop body has left A x of the unchanged candidates in Y: AMS:295 / 264)
    if (matvec_missing_into_Y(c, slots, count)) return -1;
    av_drop_all(c);
    av_mark(c, slots, count);                            // Y = A X of these rows stands until somebody writes X or Y
    { ProfScope ps(c, KC_VEC, 0, 32.0 * count * c->rows);
      maus_launch_rayleigh_dots(c->st, c->X, c->Y, c->ldp, c->d_slots, count, c->rows, c->d_c1, c->d_c2); }
    if (maus_d2h(c, num, c->d_c1, sizeof(c128) * count, c->st)) return -1;
    if (maus_d2h(c, den, c->d_c2, sizeof(c128) * count, c->st)) return -1;
    HIPCHK(c, hipStreamSynchronize(c->st));
    return 0;
}

// LU workspace for `want` simultaneous n x n systems.  Sized once (callers announce their population with maus_lu_reserve),
// grown at most once more -- then to the limit -- and never shrunk: a hipFree + hipMalloc of
// ~100 GB costs seconds, so a workspace that follows the batch size step by step (the eig population grows by up to 15
// candidates per iteration, AMS:533-534) would re-allocate inside somebody's timed region.  Callers that know their
// population announce it with maus_lu_reserve(); batches beyond the workspace run in chunks.
static int ensure_lu_ws(maus_ctx* c, int n, int want) {
    int npad = round_up(n, 32);
    if (npad > maus_lu_max_npad()) FAIL(c, "direct LU path supports n <= 8192 in this build");
    size_t per = 2 * sizeof(c128) * (size_t)npad * lu_ntiles(npad) * LU_TW;    // H and the logical-order U array (implicit pivoting), tile-major (luws.h)
    if (c->H && c->Hnpad == npad && (c->Hg >= want || c->ws_at_limit)) return 0;     // at the limit: callers chunk
    const bool second = c->H && c->Hnpad == npad;
    size_t fr = 0, tot = 0;
    HIPCHK(c, hipMemGetInfo(&fr, &tot));
    if (c->H) fr += c->Hbytes;
    int gmax = (int)std::max<size_t>(1, (size_t)(fr * 0.80) / per);
    const char* env = getenv("MAUS_LU_BATCH");
    int cap = env ? std::max(1, atoi(env)) : 512;
    int G = round_up(want, 32);
    if (c->H && c->Hnpad == npad) G = std::max(G, cap);               // a second allocation goes straight to the limit: never a third
    G = std::min(std::min(G, cap), gmax);
    if (c->H && c->Hnpad == npad && c->Hg >= G) { c->ws_at_limit = true; return 0; }   // nothing more to be had
    HIPCHK(c, hipStreamSynchronize(c->st));
    for (auto st : c->lu_st) HIPCHK(c, hipStreamSynchronize(st));
    if (c->H) { (void)hipFree(c->H); c->H = nullptr; }
    if (c->ipiv) { (void)hipFree(c->ipiv); c->ipiv = nullptr; }
    if (c->perm) { (void)hipFree(c->perm); c->perm = nullptr; }
    if (c->ident) { (void)hipFree(c->ident); c->ident = nullptr; }
    if (c->mw_sync) { (void)hipFree(c->mw_sync); c->mw_sync = nullptr; }
    if (c->info) { (void)hipFree(c->info); c->info = nullptr; }
    if (c->flags) { (void)hipFree(c->flags); c->flags = nullptr; }
    c->Hg = 0; c->Hbytes = 0; c->ws_at_limit = false;
    const int G_asked = G;
    // The big allocation can fail although hipMemGetInfo just reported the room (the 80 % rule above is a guess at what
    // fragmentation and other contexts leave).  Then make do with less -- batches beyond the workspace run in chunks -- rather
    // than fail the step.  (Round 2 also slept and retried here, on the theory that memory of a process that has just exited
    // comes back late; that was never shown and is gone.)
    {
        const int floor_g = std::min(G, std::max(32, round_up(std::min(want, 64), 32)));
        while (hipMalloc((void**)&c->H, per * G) != hipSuccess) {
            (void)hipGetLastError();
            c->H = nullptr;
            if (G > floor_g) G = std::max(floor_g, std::min(G - 32, (G * 3 / 4) / 32 * 32));
            else FAIL(c, "LU workspace: hipMalloc failed even for the smallest batch (out of device memory)");
        }
    }
    HIPCHK(c, hipMalloc((void**)&c->ipiv, sizeof(int) * (size_t)G * npad));
    HIPCHK(c, hipMalloc((void**)&c->perm, sizeof(int) * (size_t)G * npad));
    { std::vector<int> id(npad); for (int i = 0; i < npad; ++i) id[i] = i;
      HIPCHK(c, hipMalloc((void**)&c->ident, sizeof(int) * (size_t)npad));
      HIPCHK(c, hipMemcpy(c->ident, id.data(), sizeof(int) * (size_t)npad, hipMemcpyHostToDevice)); }
    HIPCHK(c, hipMalloc((void**)&c->mw_sync, maus_lu_mw_sync_bytes() * (size_t)G));
    HIPCHK(c, hipMalloc((void**)&c->info, sizeof(int) * G));
    HIPCHK(c, hipMalloc((void**)&c->flags, sizeof(int) * G));
    c->Hbytes = per * G; c->Hg = G; c->Hnpad = npad;
    c->ws_allocs++;
    // A second allocation for this matrix size, one that hit the cap, or one that had to make do with less than it asked
    // for is final: hipMemGetInfo over-reports right after another process has exited, so asking again would free and
    // re-map the whole workspace (seconds, inside somebody's step) for nothing.  Larger batches run in chunks.
    c->ws_at_limit = second || G < G_asked || G >= std::min(cap, gmax);
    return 0;
}

static LuWs make_ws(maus_ctx* c, int n, int G) {
    LuWs w;
    w.n = n; w.npad = c->Hnpad; w.ldh = w.npad + 32; w.strideH = (long)w.npad * lu_ntiles(w.npad) * LU_TW; w.G = G;
    w.H = c->H; w.U = c->H + (size_t)c->Hg * w.strideH; w.perm = c->perm; w.ident = c->ident; w.ipiv = c->ipiv; w.info = c->info; w.flags = c->flags; w.st = c->st;
    w.tick = prof_tick; w.ud = c;
    return w;
}

// Sub-batches on their own streams, each at least 64 matrices (see maus_shifted_lu_solve): MAUS_LU_STREAMS if set;
// otherwise one stream, except two for more matrices than CUs at n <= 1024 -- there the register-resident panel, the triangular
// solves and the back-substitution run one workgroup per matrix on a whole CU, so 271 matrices mean a second, nearly empty round
// of every such launch (n = 1024: 256 -> 271 matrices cost 27.8 -> 32.2 ms), and two halves of <= 256 side by side do not
// (29.2 ms; BASELINE configs[1] 9 202 -> 9 647 candidate-steps/s, profiles/r03_streams_c2.txt).  At n = 4096 the halves lose
// more on the zgemm than the tail costs (2 %).
static int lu_stream_count(int G, int npad) {
    const char* e = getenv("MAUS_LU_STREAMS");                       // read per call: tests switch it within one process
    if (e && atoi(e) > 0) return std::min(8, atoi(e));
    static const int ncu = [] { int v = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev); return v > 0 ? v : 256; }();
    return (G > ncu && npad <= 1024) ? 2 : 1;
}

static int ensure_lu_streams(maus_ctx* c, int n) {
    if (!c->ev_stage) HIPCHK(c, hipEventCreateWithFlags(&c->ev_stage, hipEventDisableTiming));
    while ((int)c->lu_st.size() < n) {
        hipStream_t s; hipEvent_t e;
        // descending priorities keep the sub-batches out of phase: while the first one is in an MFMA-bound trailing
        // update the next one gets the left-over issue slots for its bandwidth-bound panel work, and vice versa
        int plo = 0, phi = 0;
        (void)hipDeviceGetStreamPriorityRange(&plo, &phi);          // plo = least, phi = greatest (numerically lower)
        const int idx = (int)c->lu_st.size();
        const int prio = std::max(phi, std::min(plo, phi + idx));
        HIPCHK(c, hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio));
        HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(e, s));                            // the hardware queue behind a stream is set up at its first
        HIPCHK(c, hipEventSynchronize(e));                          // command: here, not inside the first batch that uses it
        c->lu_st.push_back(s); c->lu_done.push_back(e);
    }
    return 0;
}

static int lu_nbo() { const char* e = getenv("MAUS_LU_NBO"); int v = e ? atoi(e) : 512; if (v < 32) v = 32; return (v / 32) * 32; }

// info < 0 is never a numerical outcome: INT_MIN is written by the multi-workgroup panel when a rendezvous between the
// workgroups of one matrix timed out (lu.hip) -- that matrix is then half factored and its "solution" is garbage.
static bool any_internal_failure(int G, const int* info) {
    for (int g = 0; g < G; ++g) if (info[g] < 0) return true;
    return false;
}

static void finish_status(int G, const int* info, const int* flags, int32_t* status) {
    for (int g = 0; g < G; ++g) {
        if (flags[g] & 1) status[g] = -1;
        else if (info[g] > 0) status[g] = info[g];
        else if (flags[g] & 2) status[g] = -2;
        else status[g] = 0;
    }
}

static bool mw_allowed(const maus_ctx* c) { return !c->shared_device && !c->mw_disabled; }
static void mw_configure(const maus_ctx* c, LuWs& w) {
    w.mw_sync = mw_allowed(c) ? c->mw_sync : nullptr;
    const char* e = getenv("MAUS_PANEL_MW_FORCE_ABORT");          // test hook: see lu_panel_mw_kernel
    if (e && atoi(e)) { w.mw_force_abort = 1; w.mw_timeout = 2000000ull; }   // 20 ms instead of 2 s
}


// generator-state and index buffers of sub-batch `sbi` (also called from maus_lu_reserve, so that the first batch that runs
// as two sub-batches does not allocate inside somebody's timed step)
static int mt_buf_reserve(maus_ctx* c, int sbi, int ngen, size_t nints) {
    if ((int)c->mt_bufs.size() <= sbi) c->mt_bufs.resize(sbi + 1);
    maus_ctx::MtBuf& mb = c->mt_bufs[sbi];
    if (ngen > mb.cap) {
        void** ps[] = {(void**)&mb.states, (void**)&mb.base};
        for (auto p : ps) if (*p) { (void)hipFree(*p); *p = nullptr; }
        const int cap = std::max(ngen, 4096);
        HIPCHK(c, hipMalloc((void**)&mb.states, sizeof(uint32_t) * 624 * (size_t)cap));
        HIPCHK(c, hipMalloc((void**)&mb.base, sizeof(uint32_t) * 624));
        mb.cap = cap;
    }
    if (nints > mb.int_cap) {
        if (mb.ints) { (void)hipFree(mb.ints); mb.ints = nullptr; }
        const size_t cap = std::max<size_t>(nints * 2, 65536);
        HIPCHK(c, hipMalloc((void**)&mb.ints, sizeof(int) * cap));
        mb.int_cap = cap;
    }
    return 0;
}

// MAUS_PERT_MT19937: generator start states for candidates [first, first+g) of the run by binary lifting (over the
// draw index m, then over the sub-stream index b), then the H build that regenerates the draws (mtdev.hip).  The plan
// itself (pure host arithmetic on stream offsets) lives in mtplan.cpp so that it can be built and run under the CPU
// sanitizers (`make asan`).
static int mt_prepare_and_build(maus_ctx* c, const LuWs& w, const maus_mt_desc* d, int first, int g, int rhs_mode, int lo, int sbi, int tiled) {
    if ((int)c->mt_bufs.size() <= sbi) c->mt_bufs.resize(sbi + 1);
    maus_ctx::MtBuf& mb = c->mt_bufs[sbi];
    const int n = w.n;
    int s_override = 0;
    { const char* e = getenv("MAUS_MT_SUBSTREAMS"); if (e && atoi(e) > 0) s_override = std::min(64, atoi(e)); }
    MausMtPlan& pl = c->mt_plan;
    const char* perr = nullptr;
    if (maus_mt_plan(d, n, first, g, s_override, &pl, &perr)) FAIL(c, perr ? perr : "maus_mt_plan failed");
    const int ngen = pl.ngen;
    if (mt_buf_reserve(c, sbi, ngen, pl.hs.size())) return -1;
    struct DevLevel { size_t off; int count; MausJumpPolys P; int src_off; bool multi; };
    std::vector<DevLevel> levels;
    auto poly_taps = [&](uint64_t J, MausJumpPolys& P, int v) -> int {     // tap list of x^J mod phi, cached on the device
        auto it = c->mt_taps.find(J);
        if (it == c->mt_taps.end()) {
            std::vector<uint64_t> poly(312);
            if (maus_mt_jump_poly(J, poly.data())) FAIL(c, "MT19937 jump polynomial failed");
            const int split = maus_mt_tap_split(), ztap = maus_mt_zero_tap();
            std::vector<int> taps;
            int nlo16 = 0;
            for (int half = 0; half < 2; ++half) {
                for (int i = half ? split : 0; i < (half ? 19937 : split); ++i)
                    if ((poly[i >> 6] >> (i & 63)) & 1ull) taps.push_back(half ? i - split : i);
                while (taps.size() % 16) taps.push_back(ztap);
                if (!half) nlo16 = (int)(taps.size() / 16);
            }
            if (taps.empty()) taps.assign(16, ztap);
            int* dt = nullptr;
            HIPCHK(c, hipMalloc((void**)&dt, sizeof(int) * taps.size()));
            HIPCHK(c, hipMemcpy(dt, taps.data(), sizeof(int) * taps.size(), hipMemcpyHostToDevice));
            it = c->mt_taps.emplace(J, maus_ctx::MtTaps{dt, (int)(taps.size() / 16), nlo16}).first;
        }
        P.taps[v] = it->second.taps; P.ntap16[v] = it->second.ntap16; P.nlo16[v] = it->second.nlo16;
        return 0;
    };
    for (const MausMtPlan::Level& L : pl.levels) {
        DevLevel D{L.off, L.count, {}, L.src_off, L.multi};
        memset(&D.P, 0, sizeof D.P);
        bool need[16] = {false};
        if (L.multi) for (int i = 0; i < L.count; ++i) need[pl.hs[L.off + L.count + i] & 15] = true;
        else need[1] = true;
        for (int v = 1; v < 16; ++v) if (need[v] && poly_taps((uint64_t)v * L.J, D.P, v)) return -1;
        levels.push_back(D);
    }
    const std::vector<int>& hs = pl.hs;
    if (maus_h2d(c, mb.base, d->key, sizeof(uint32_t) * 624, w.st)) return -1;
    if (maus_h2d(c, mb.ints, hs.data(), sizeof(int) * hs.size(), w.st)) return -1;
    HIPCHK(c, hipStreamSynchronize(w.st));                       // staging (hs, d->key) is reusable from here on
    maus_mt_copy_states(w.st, mb.states, mb.base, ngen);
    for (const DevLevel& L : levels) maus_mt_jump(w.st, mb.states, mb.ints + L.off, L.multi ? mb.ints + L.off + L.count : nullptr, L.count, L.P, L.src_off);
    const int* d_extra = mb.ints; const int* d_rpos = mb.ints + ngen;
    prof_tick(c, KC_BUILD, 0, 0, 0);
    maus_build_h_mt(w.st, c->A, n, w.npad, w.ldh, w.strideH, w.H, g, pl.S, (long)pl.E, c->d_c1 + lo, c->d_r1 + lo, rhs_mode, c->X, c->ldp,
                    c->d_slots + lo, c->b, mb.states, d_extra, d_rpos, w.flags, tiled);
    prof_tick(c, KC_BUILD, 1, 0, 32.0 * w.npad * w.ldh * g);
    return 0;
}

int maus_shifted_lu_solve(maus_ctx* c, const int* slots, int count, const double* shift, const double* psi,
                          int rhs_mode, int pert_mode, const void* pert_data, int32_t* status) {
    av_drop_all(c);
    if (!c->A || !c->X) FAIL(c, "maus_shifted_lu_solve: matrix/population missing");
    if (c->rows != c->cols) FAIL(c, "maus_shifted_lu_solve: square matrix required");
    if (rhs_mode == 1 && (!c->b || c->bn != c->rows)) FAIL(c, "maus_shifted_lu_solve: rhs b not set");
    if ((pert_mode == MAUS_PERT_UNIFORM || pert_mode == MAUS_PERT_MT19937) && !pert_data) FAIL(c, "pert_data missing");
    if (count == 0) return 0;
    const int n = c->rows;
    if (check_slots(c, slots, count)) return -1;
    if (ensure_scalars(c, count)) return -1;
    if (ensure_lu_ws(c, n, count)) return -1;
    // balanced chunks (271 candidates in a 256-matrix workspace run as 136 + 135, not 256 + 15)
    const int nchunks = (count + c->Hg - 1) / c->Hg;
    const int Gmax = (count + nchunks - 1) / nchunks;
    const int nst = lu_stream_count(Gmax, c->Hnpad);
    if (ensure_lu_streams(c, nst)) return -1;
    std::vector<int> h_info(Gmax), h_flags(Gmax);
    for (int off = 0; off < count; off += Gmax) {
        const int G = std::min(Gmax, count - off);
        if (maus_h2d(c, c->d_slots, slots + off, sizeof(int) * G, c->st)) return -1;
        if (maus_h2d(c, c->d_c1, shift + 2 * (size_t)off, sizeof(c128) * G, c->st)) return -1;
        if (maus_h2d(c, c->d_r1, psi + off, sizeof(double) * G, c->st)) return -1;
        const double* dU = nullptr;
        if (pert_mode == MAUS_PERT_UNIFORM) {
            size_t ub = sizeof(double) * 2 * (size_t)n * n * G;
            if (ub > c->Ubytes) { if (c->Upert) (void)hipFree(c->Upert); c->Upert = nullptr; HIPCHK(c, hipMalloc((void**)&c->Upert, ub)); c->Ubytes = ub; }
            if (maus_stage_h2d(c, c->Upert, (const double*)pert_data + 2 * (size_t)n * n * off, ub, c->st)) return -1;
            dU = c->Upert;
        }
        // One stream at n = 4096 (lu_stream_count).  MAUS_LU_STREAMS=<n> splits a batch into min(n, G / 64)
        // sub-batches on their own streams so that the bandwidth- and latency-bound phases of one (panel, triangular solves,
        // H build) run beside the MFMA-bound trailing updates of the others.  That paid 0-2 % while those phases ran at
        // 2.4-4 TB/s (round 2, where a run-time tuner picked the count per batch-size class); with the tile-major workspace
        // they run at ~5 TB/s, the zgemm beside them loses more than they gain, and whole driver-shaped runs give 352 / 338
        // candidate-steps/s with 1 / 2 streams (profiles/r03_streams_fixed.txt) -- the tuner, which picked three from one
        // noisy sample per count, is gone.  Results do not depend on the split (tests/test_gpu_bench_path.py: bit-equal).
        constexpr int min_sub = 64;                 // smallest sub-batch that gets a stream of its own (round 4, 16 / 32 / 64 solves split into sub-batches of 8-32 on 2-4 streams: 3-30 % slower, profiles/r04_small_batch_streams_negative.txt)
        const int S = std::max(1, std::min(nst, G / min_sub));
        // One more pass without the multi-workgroup panel if a rendezvous of it timed out (info = INT_MIN): its premise --
        // all workgroups of a matrix resident at once -- does not hold on a device that somebody else is using too.  The
        // pass rebuilds H from the same inputs, so the results are those of a context that never used that kernel.
        for (int pass = 0; pass < 2; ++pass) {
        HIPCHK(c, hipMemsetAsync(c->info, 0, sizeof(int) * G, c->st));
        HIPCHK(c, hipMemsetAsync(c->flags, 0, sizeof(int) * G, c->st));
        if (S > 1) HIPCHK(c, hipEventRecord(c->ev_stage, c->st));
        std::vector<LuWs> wss;
        std::vector<int> los;
        for (int sb = 0; sb < S; ++sb) {
            const int lo = (int)((long)G * sb / S), hi = (int)((long)G * (sb + 1) / S), g = hi - lo;
            if (g <= 0) continue;
            hipStream_t st = (S == 1) ? c->st : c->lu_st[sb];
            if (S > 1) HIPCHK(c, hipStreamWaitEvent(st, c->ev_stage, 0));
            LuWs w = make_ws(c, n, g);
            w.H += (long)lo * w.strideH; w.U += (long)lo * w.strideH; w.perm += (long)lo * w.npad; w.ipiv += (long)lo * w.npad; w.info += lo; w.flags += lo; w.st = st;
            if (S == 1) mw_configure(c, w);              // the only LU in flight on this device: the panel may spread over several workgroups per matrix
            c->prof_st = st;
            if (pert_mode == MAUS_PERT_MT19937) {
                if (mt_prepare_and_build(c, w, (const maus_mt_desc*)pert_data, off + lo, g, rhs_mode, lo, sb, 1)) return -1;
            } else
            maus_build_h(w, c->A, c->d_c1 + lo, c->d_r1 + lo, rhs_mode, c->X, c->ldp, c->d_slots + lo, c->b, pert_mode,
                         dU ? dU + 2 * (size_t)n * n * lo : nullptr, 1);
            wss.push_back(w); los.push_back(lo);
        }
        // (A token schedule that lets only one sub-batch at a time into its panel phase -- so that it always runs
        // beside the other's trailing update -- was measured and is slower, 265 vs 276 steps/s: the panel kernel
        // needs whole CUs, cannot co-run with a saturating zgemm, and the forced alternation only adds waits.)
        for (auto& w : wss) { c->prof_st = w.st; maus_lu_factor(w, lu_nbo()); }
        for (size_t i = 0; i < wss.size(); ++i) {
            c->prof_st = wss[i].st;
            maus_lu_backsolve(wss[i], c->W, c->ldp, c->d_slots + los[i], nullptr);
            if (S > 1) { HIPCHK(c, hipEventRecord(c->lu_done[i], wss[i].st)); HIPCHK(c, hipStreamWaitEvent(c->st, c->lu_done[i], 0)); }
        }
        c->prof_st = nullptr;
        if (maus_d2h(c, h_info.data(), c->info, sizeof(int) * G, c->st)) return -1;
        if (maus_d2h(c, h_flags.data(), c->flags, sizeof(int) * G, c->st)) return -1;
        HIPCHK(c, hipStreamSynchronize(c->st));
        HIPCHK(c, hipGetLastError());
        if (any_internal_failure(G, h_info.data())) {
            if (pass == 0 && !c->mw_disabled) { c->mw_disabled = true; c->mw_aborts++; continue; }
            FAIL(c, "maus_shifted_lu_solve: internal error: LU panel rendezvous timed out (info < 0) and the batch could not be repeated");
        }
        break;
        }   // pass
        finish_status(G, h_info.data(), h_flags.data(), status + off);
    }
    return 0;
}

int maus_lu_solve_host(maus_ctx* c, int count, int n, const double* a, const double* b, double* x, int32_t* status, int32_t* ipiv_out) {
    if (count <= 0 || n <= 0) FAIL(c, "maus_lu_solve_host: bad sizes");
    if (ensure_lu_ws(c, n, count)) return -1;
    const int Gmax = c->Hg;
    size_t ab = sizeof(c128) * (size_t)n * n, bb = sizeof(c128) * (size_t)n;
    if (ensure_scratch(c, (ab + 2 * bb) * Gmax)) return -1;
    c128* dA = (c128*)c->scratch; c128* dB = dA + (size_t)n * n * Gmax; c128* dX = dB + (size_t)n * Gmax;
    std::vector<int> h_info(Gmax), h_flags(Gmax);
    for (int off = 0; off < count; off += Gmax) {
        const int G = std::min(Gmax, count - off);
        if (maus_stage_h2d(c, dA, a + 2 * (size_t)n * n * off, ab * G, c->st)) return -1;
        if (maus_stage_h2d(c, dB, b + 2 * (size_t)n * off, bb * G, c->st)) return -1;
        for (int pass = 0; pass < 2; ++pass) {
        LuWs w = make_ws(c, n, G);
        mw_configure(c, w);
        HIPCHK(c, hipMemsetAsync(c->info, 0, sizeof(int) * G, c->st));
        HIPCHK(c, hipMemsetAsync(c->flags, 0, sizeof(int) * G, c->st));
        maus_load_h(w, dA, dB);
        maus_lu_factor(w, lu_nbo());
        maus_lu_backsolve(w, nullptr, 0, nullptr, dX);
        if (maus_stage_d2h(c, x + 2 * (size_t)n * off, dX, bb * G, c->st)) return -1;
        if (maus_d2h(c, h_info.data(), c->info, sizeof(int) * G, c->st)) return -1;
        if (maus_d2h(c, h_flags.data(), c->flags, sizeof(int) * G, c->st)) return -1;
        if (ipiv_out) {
            // ipiv rows are npad long on the device; return the first n of each
            HIPCHK(c, hipMemcpy2DAsync(ipiv_out + (size_t)off * n, sizeof(int) * n, c->ipiv, sizeof(int) * c->Hnpad, sizeof(int) * n, G, hipMemcpyDeviceToHost, c->st));
        }
        HIPCHK(c, hipStreamSynchronize(c->st));
        HIPCHK(c, hipGetLastError());
        if (any_internal_failure(G, h_info.data())) {                 // see maus_shifted_lu_solve
            if (pass == 0 && !c->mw_disabled) { c->mw_disabled = true; c->mw_aborts++; continue; }
            FAIL(c, "maus_lu_solve_host: internal error: LU panel rendezvous timed out (info < 0) and the batch could not be repeated");
        }
        break;
        }   // pass
        finish_status(G, h_info.data(), h_flags.data(), status + off);
    }
    return 0;
}

int maus_lu_reserve(maus_ctx* c, int n, int count, int* capacity_out) {
    if (n <= 0 || count < 0) FAIL(c, "maus_lu_reserve: bad sizes");
    if (count > 0 && ensure_lu_ws(c, n, count)) return -1;
    if (count > 0 && c->H) {                                       // what a batch of the announced size will use (lu_stream_count)
        const int nst = lu_stream_count(std::min(count, c->Hg), c->Hnpad);
        if (ensure_lu_streams(c, nst)) return -1;
        for (int sb = 0; sb < nst; ++sb) if (mt_buf_reserve(c, sb, 4096, 65536)) return -1;
    }
    if (capacity_out) *capacity_out = (c->H && c->Hnpad == round_up(n, 32)) ? c->Hg : 0;
    return 0;
}

int maus_lu_workspace_allocs(maus_ctx* c) { return c ? c->ws_allocs : -1; }

int maus_set_shared_device(maus_ctx* c, int shared) { if (!c) return -1; c->shared_device = shared != 0; return 0; }

int maus_lu_mw_aborts(maus_ctx* c) { return c ? c->mw_aborts : -1; }

int maus_relax_normalise(maus_ctx* c, const int* slots, int count, const double* alpha, int normalise, double* norm_out) {
    av_drop_all(c);
    if (!c->X) FAIL(c, "maus_relax_normalise: population missing");
    if (count == 0) return 0;
    if (upload_slots(c, slots, count)) return -1;
    const int n = c->rows;
    if (maus_h2d(c, c->d_c1, alpha, sizeof(c128) * count, c->st)) return -1;
    { ProfScope ps(c, KC_VEC, 0, 48.0 * count * n);
      maus_launch_relax(c->st, c->X, c->W, c->ldp, c->d_slots, count, n, c->d_c1, normalise, c->d_r1); }
    if (maus_d2h(c, norm_out, c->d_r1, sizeof(double) * count, c->st)) return -1;
    HIPCHK(c, hipStreamSynchronize(c->st));
    return 0;
}

int maus_residual(maus_ctx* c, int kind, const int* slots, int count, const double* lam, double* resid, int32_t* finite) {
    // (SVD) does S hold A^H u for every row asked for?  Decided before the stamps go: this call rewrites Y
    bool have_ahu = (kind == MAUS_SVD) && av_family(c, count, c->rows);
    for (int k = 0; have_ahu && k < count; ++k) have_ahu = slots[k] >= 0 && ahu_has(c, slots[k]);
    av_drop_all(c);
    if (!c->A || !c->X) FAIL(c, "maus_residual: matrix/population missing");
    if (count == 0) return 0;
    if (upload_slots(c, slots, count)) return -1;
    if (kind == MAUS_EIG || kind == MAUS_LINEAR) {
        if (c->rows != c->cols) FAIL(c, "maus_residual: square matrix required");
        if (kind == MAUS_LINEAR && (!c->b || c->bn != c->rows)) FAIL(c, "maus_residual: rhs b not set");
        if (kind == MAUS_EIG && !lam) FAIL(c, "maus_residual: lambda missing");
        if (lam) { if (maus_h2d(c, c->d_c1, lam, sizeof(c128) * count, c->st)) return -1; }
        matvec_into_Y(c, c->X, count);
        { ProfScope ps(c, KC_VEC, 0, 32.0 * count * c->rows);
          maus_launch_residual(c->st, kind, c->X, c->Y, c->ldp, c->d_slots, count, c->rows, lam ? c->d_c1 : nullptr, c->b, c->d_r1, c->d_i1); }
        av_mark(c, slots, count);                        // Y = A X of these rows stands until somebody writes X or Y
    } else if (kind == MAUS_SVD) {
        if (!lam) FAIL(c, "maus_residual: sigma missing");
        if (maus_h2d(c, c->d_c1, lam, sizeof(c128) * count, c->st)) return -1;
        // ||A v - s u||  : Y = X * A^T  (count x rows)
        matvec_into_Y(c, c->X, count);
        maus_launch_svd_resid(c->st, c->Y, c->U, c->ldp, c->d_slots, count, c->rows, c->d_c1, c->d_r1, 0, c->d_i1);
        // ||A^H u - s v||: W = U * conj(A)  (count x cols), B = conj(A) as [k=rows][n=cols] -- or, when the power step of this
        // loop body has left exactly that product in S (same u, same kernel, same bits), no product at all (round 4)
        if (!have_ahu) {
            ProfScope ps(c, KC_GEMM, 8.0 * count * c->rows * c->cols, 16.0 * (double)c->rows * c->cols);
            maus_zgemm_launch_idx(c->st, count, c->cols, c->rows, c->U, c->ldp, 0, c->A, c->cols, 0, c->W, c->ldp, 0,
                                  1.0, 0, 1, 0, false, true, c->d_slots, c->d_slots);
        }
        maus_launch_svd_resid(c->st, have_ahu ? c->S : c->W, c->X, c->ldp, c->d_slots, count, c->cols, c->d_c1, c->d_r1, 1, c->d_i1);
        av_mark(c, slots, count);                        // Y = A X of these rows stands until somebody writes X or Y
    } else FAIL(c, "maus_residual: unknown kind");
    if (maus_d2h(c, resid, c->d_r1, sizeof(double) * count, c->st)) return -1;
    if (maus_d2h(c, finite, c->d_i1, sizeof(int) * count, c->st)) return -1;
    HIPCHK(c, hipStreamSynchronize(c->st));
    return 0;
}

// The step in two halves: propose leaves X / U alone (u_new in Y, v_new in W), commit copies them in for the candidates
// the host accepts.  A collapse (AMS:229-232, 236-239) ends the speculative run at that candidate and the ones behind it
// must be stepped again from their own vectors: with the proposal held back nothing has to be restored -- until round 3
// the engine kept a host copy of every candidate's vectors for that, 400 MB over PCIe per loop body at BASELINE
// configs[4] (6144 candidates x 2048), 57 % of its wall time.
int maus_svd_power_propose(maus_ctx* c, const int* slots, int count, double* norms_out) {
    if (!c->A || !c->X) FAIL(c, "maus_svd_power_propose: matrix/population missing");
    if (count == 0) return 0;
    if (upload_slots(c, slots, count)) return -1;
    // ||v_in||
    maus_launch_norm(c->st, c->X, c->ldp, c->d_slots, count, c->cols, c->d_r1, 4, 0);
    // t = A v -> Y ; sigma1 = ||t|| ; u = t / (sigma1 > 1e-10 ? sigma1 : 1), in place.
    // Round 4 (later): the rows whose Y the residual of the previous loop body left behind are not multiplied again
    // (matvec_missing_into_Y) -- at BASELINE configs[4] that is all but the ~15 spawns of a body, one of the four products of a loop body.
    if (matvec_missing_into_Y(c, slots, count)) return -1;
    av_drop_all(c);                                      // Y becomes u below
    maus_launch_norm_scale(c->st, c->Y, c->Y, c->ldp, c->d_slots, count, c->rows, c->d_r1, 4, 1);
    maus_launch_norm(c->st, c->Y, c->ldp, c->d_slots, count, c->rows, c->d_r1, 4, 2);
    // s = A^H u -> W ; sigma2 = ||s|| ; v = s / (sigma2 > 1e-10 ? sigma2 : 1), in place
    // (the unscaled product stays in S: the residual of this loop body asks for A^H u of the same u, AMS:298)
    if (c->Scap < c->cap) {
        HIPCHK(c, hipStreamSynchronize(c->st));
        if (c->S) { (void)hipFree(c->S); c->S = nullptr; c->Scap = 0; }
        HIPCHK(c, hipMalloc((void**)&c->S, sizeof(c128) * (size_t)c->cap * c->ldp));
        c->Scap = c->cap;
    }
    { ProfScope ps(c, KC_GEMM, 8.0 * count * c->rows * c->cols, 16.0 * (double)c->rows * c->cols);
      maus_zgemm_launch_idx(c->st, count, c->cols, c->rows, c->Y, c->ldp, 0, c->A, c->cols, 0, c->S, c->ldp, 0,
                            1.0, 0, 1, 0, false, true, c->d_slots, c->d_slots); }
    maus_launch_norm_scale(c->st, c->S, c->W, c->ldp, c->d_slots, count, c->cols, c->d_r1, 4, 3);
    if (++c->prop_epoch == 0) { c->prop_epoch = 1; std::fill(c->prop_stamp.begin(), c->prop_stamp.end(), 0u); }
    if (c->prop_stamp.size() < (size_t)c->cap) c->prop_stamp.resize(c->cap, 0u);
    if (av_family(c, count, c->rows)) for (int k = 0; k < count; ++k) c->prop_stamp[slots[k]] = c->prop_epoch;
    if (maus_d2h(c, norms_out, c->d_r1, sizeof(double) * 4 * count, c->st)) return -1;
    HIPCHK(c, hipStreamSynchronize(c->st));
    return 0;
}

int maus_svd_commit(maus_ctx* c, const int* slots, int count) {
    if (++c->av_epoch == 0) { c->av_epoch = 1; std::fill(c->av_stamp.begin(), c->av_stamp.end(), 0u); }     // X changes; S stays what it is
    if (!c->X) FAIL(c, "maus_svd_commit: population missing");
    if (count == 0) return 0;
    if (upload_slots(c, slots, count)) return -1;
    hipLaunchKernelGGL(copy_rows_kernel, dim3(count), dim3(256), 0, c->st, c->U, c->Y, c->ldp, c->d_slots, c->rows);
    hipLaunchKernelGGL(copy_rows_kernel, dim3(count), dim3(256), 0, c->st, c->X, c->W, c->ldp, c->d_slots, c->cols);
    HIPCHK(c, hipStreamSynchronize(c->st));
    // U is now the u whose A^H u the latest proposal left in S -- for the rows that proposal covered
    if (c->ahu_stamp.size() < (size_t)c->cap) c->ahu_stamp.resize(c->cap, 0u);
    for (int k = 0; k < count; ++k)
        if ((size_t)slots[k] < c->prop_stamp.size() && c->prop_stamp[slots[k]] == c->prop_epoch) c->ahu_stamp[slots[k]] = c->ahu_epoch;
    // (a proposal of up to 32 rows left no prop stamps: see av_family)
    return 0;
}

int maus_svd_power_step(maus_ctx* c, const int* slots, int count, double* norms_out) {
    if (maus_svd_power_propose(c, slots, count, norms_out)) return -1;
    return maus_svd_commit(c, slots, count);
}

int maus_herm_match(maus_ctx* c, const int* slots, int count, int32_t* idx_out, double* norm_out) {
    av_drop_all(c);
    if (!c->V || !c->X) FAIL(c, "maus_herm_match: eigenvectors/population missing");
    if (count == 0) return 0;
    if (upload_slots(c, slots, count)) return -1;
    const int n = c->rows;
    // scores: Y[slot][j] = sum_i conj(X[slot][i]) V[i][j]
    { ProfScope ps(c, KC_GEMM, 8.0 * count * n * n, 16.0 * (double)n * n);
      maus_zgemm_launch_idx(c->st, count, n, n, c->X, c->ldp, 0, c->V, n, 0, c->Y, c->ldp, 0, 1.0, 0, 1, 0, true, false, c->d_slots, c->d_slots); }
    maus_launch_herm_pick(c->st, c->Y, c->ldp, c->X, c->ldp, c->d_slots, count, c->V, n, c->d_i1, c->d_r1);
    if (maus_d2h(c, idx_out, c->d_i1, sizeof(int) * count, c->st)) return -1;
    if (maus_d2h(c, norm_out, c->d_r1, sizeof(double) * count, c->st)) return -1;
    HIPCHK(c, hipStreamSynchronize(c->st));
    return 0;
}

// Gram block of candidate vectors (SURVEY f-2): G[i][j] = vdot(x_i, x_j) = sum_k conj(x_i[k]) x_j[k] over the rows
// `slots` of population array `which`.  The rows are gathered into scratch once; the product is one zgemm in
// dot-product layout with a conjugated left operand -- the same kernel as the Hermitian similarity row.

int maus_gram(maus_ctx* c, int which, const int* slots, int count, int len, double* out_c128) {
    c128* X = (which == MAUS_POP_X) ? c->X : (which == MAUS_POP_U) ? c->U : (which == MAUS_POP_W) ? c->W : nullptr;
    if (!X) FAIL(c, "maus_gram: population array missing");
    if (len <= 0 || len > c->ldp) FAIL(c, "maus_gram: bad vector length");
    if (count == 0) return 0;
    if (upload_slots(c, slots, count)) return -1;
    const size_t rows = (size_t)count * len, g = (size_t)count * count;
    if (ensure_scratch(c, sizeof(c128) * (rows + g))) return -1;
    c128* R = (c128*)c->scratch; c128* G = R + rows;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(count), dim3(256), 0, c->st, X, c->ldp, c->d_slots, len, R);
    { ProfScope ps(c, KC_GEMM, 8.0 * count * count * len, 16.0 * (2.0 * rows + g));
      maus_zgemm_launch_idx(c->st, count, count, len, R, len, 0, R, len, 0, G, count, 0, 1.0, 0, 1, 1, true, false, nullptr, nullptr); }
    if (maus_stage_d2h(c, out_c128, G, sizeof(c128) * g, c->st)) return -1;
    HIPCHK(c, hipGetLastError());
    return 0;
}

int maus_gmres(maus_ctx* c, const int* slots, int count, const double* shift, const double* psi, int rhs_mode,
               const int32_t* use_jacobi, double rtol, int restart, int maxiter, int32_t* info_out, int32_t* inner_out, int32_t* status) {
    av_drop_all(c);
    return maus_gmres_run(c, slots, count, shift, psi, rhs_mode, use_jacobi, rtol, restart, maxiter, info_out, inner_out, status,
                          nullptr, 0, 0, nullptr);
}

int maus_gmres_pert(maus_ctx* c, const int* slots, int count, const double* shift, const double* psi, int rhs_mode,
                    const int32_t* want_jacobi, int pert_mode, const void* pert_data, double rtol, int restart, int maxiter,
                    int32_t* info_out, int32_t* inner_out, int32_t* status, int32_t* jacobi_out) {
    av_drop_all(c);
    if (!c->A || !c->X) FAIL(c, "maus_gmres_pert: matrix/population missing");
    if (c->rows != c->cols) FAIL(c, "maus_gmres_pert: square matrix required");
    if (rhs_mode == 1 && (!c->b || c->bn != c->rows)) FAIL(c, "maus_gmres_pert: rhs b not set");
    if ((pert_mode == MAUS_PERT_UNIFORM || pert_mode == MAUS_PERT_MT19937) && !pert_data) FAIL(c, "pert_data missing");
    if (count == 0) return 0;
    const int n = c->rows;
    if (check_slots(c, slots, count)) return -1;
    if (ensure_scalars(c, count)) return -1;
    if (ensure_lu_ws(c, n, count)) return -1;
    std::vector<int> h_flags(std::min(count, c->Hg));
    for (int off = 0; off < count; off += c->Hg) {                     // chunks of at most the workspace capacity
        const int G = std::min(c->Hg, count - off);
        if (maus_h2d(c, c->d_slots, slots + off, sizeof(int) * G, c->st)) return -1;
        if (maus_h2d(c, c->d_c1, shift + 2 * (size_t)off, sizeof(c128) * G, c->st)) return -1;
        if (maus_h2d(c, c->d_r1, psi + off, sizeof(double) * G, c->st)) return -1;
        HIPCHK(c, hipMemsetAsync(c->flags, 0, sizeof(int) * G, c->st));
        LuWs w = make_ws(c, n, G);
        const double* dU = nullptr;
        if (pert_mode == MAUS_PERT_UNIFORM) {
            size_t ub = sizeof(double) * 2 * (size_t)n * n * G;
            if (ub > c->Ubytes) { if (c->Upert) (void)hipFree(c->Upert); c->Upert = nullptr; HIPCHK(c, hipMalloc((void**)&c->Upert, ub)); c->Ubytes = ub; }
            if (maus_stage_h2d(c, c->Upert, (const double*)pert_data + 2 * (size_t)n * n * off, ub, c->st)) return -1;
            dU = c->Upert;
        }
        if (pert_mode == MAUS_PERT_MT19937) {
            if (mt_prepare_and_build(c, w, (const maus_mt_desc*)pert_data, off, G, rhs_mode, 0, 0, 0)) return -1;
        } else
            maus_build_h(w, c->A, c->d_c1, c->d_r1, rhs_mode, c->X, c->ldp, c->d_slots, c->b, pert_mode, dU, 0);     // row-major: one dense GEMV operand per candidate
        if (maus_d2h(c, h_flags.data(), c->flags, sizeof(int) * G, c->st)) return -1;
        if (maus_gmres_run(c, slots + off, G, shift + 2 * (size_t)off, psi + off, rhs_mode, want_jacobi + off, rtol, restart, maxiter,
                           info_out + off, inner_out + off, status + off, w.H, w.ldh, w.strideH, jacobi_out ? jacobi_out + off : nullptr))
            return -1;
        for (int g = 0; g < G; ++g) if (h_flags[g] & 1) status[off + g] = -1;       // non-finite H_k / rhs (AMS:94 analogue)
    }
    return 0;
}

int maus_jacobi_check(maus_ctx* c, int count, const double* shift, const double* psi, int32_t* ok) {
    return maus_jacobi_check_run(c, count, shift, psi, ok);
}

int maus_zgemm_host(maus_ctx* c, int M, int N, int K, const double* A, const double* B, double* C, int b_layout, int conj_a, int conj_b, double alpha, int beta) {
    if (M <= 0 || N <= 0 || K <= 0) FAIL(c, "maus_zgemm_host: bad sizes");
    size_t ea = (size_t)M * K, eb = (size_t)K * N, ec = (size_t)M * N;
    if (ensure_scratch(c, sizeof(c128) * (ea + eb + ec))) return -1;
    c128* dA = (c128*)c->scratch; c128* dB = dA + ea; c128* dC = dB + eb;
    if (maus_stage_h2d(c, dA, A, sizeof(c128) * ea, c->st) || maus_stage_h2d(c, dB, B, sizeof(c128) * eb, c->st)
        || maus_stage_h2d(c, dC, C, sizeof(c128) * ec, c->st)) return -1;
    { ProfScope ps(c, KC_GEMM, 8.0 * M * N * K, 16.0 * (ea + eb + 2.0 * ec));
      maus_zgemm_launch_idx(c->st, M, N, K, dA, K, 0, dB, b_layout ? K : N, 0, dC, N, 0, alpha, beta, 1, b_layout, conj_a != 0, conj_b != 0, nullptr, nullptr); }
    if (maus_stage_d2h(c, C, dC, sizeof(c128) * ec, c->st)) return -1;
    HIPCHK(c, hipGetLastError());
    return 0;
}

int maus_timer_start(maus_ctx* c) { HIPCHK(c, hipEventRecord(c->t0, c->st)); return 0; }
int maus_timer_stop(maus_ctx* c, float* ms) {
    HIPCHK(c, hipEventRecord(c->t1, c->st));
    HIPCHK(c, hipEventSynchronize(c->t1));
    HIPCHK(c, hipEventElapsedTime(ms, c->t0, c->t1));
    return 0;
}

int maus_profile_enable(maus_ctx* c, int on) {
    prof_resolve(c);
    c->prof_on = on != 0;
    c->prof_mode = (on == 2) ? 2 : 1;
    c->prof_seq = 0;
    for (int k = 0; k < KC_COUNT; ++k) c->prof_cnt[k] = 0;
    if (const char* e = getenv("MAUS_PROF_STRIDE")) {           // "big,small"
        int a = 1, b = 0;
        if (sscanf(e, "%d,%d", &a, &b) >= 1) { c->prof_stride_big = std::max(1, a); c->prof_stride_small = std::max(0, b); }
    }
    if (on) for (int k = 0; k < KC_COUNT; ++k) { c->launches[k] = 0; c->ms[k] = 0; c->flops[k] = 0; c->bytes[k] = 0; c->total_launches[k] = 0; c->prof_iv[k].clear(); }
    if (on) {
        if (!c->prof_origin) HIPCHK(c, hipEventCreate(&c->prof_origin));
        HIPCHK(c, hipEventRecord(c->prof_origin, c->st));
        HIPCHK(c, hipStreamSynchronize(c->st));
    }
    return 0;
}

// Length (ms) of the UNION of the bracketed launches' time intervals of one class, over every stream they ran on:
// the time during which at least one such kernel was executing.  With sub-batches on two streams two trailing updates
// often run side by side and share the machine; their flops divided by this union is the rate the class achieved,
// whereas the sum of the individual durations counts the shared time twice.
int maus_profile_union_ms(maus_ctx* c, int klass, double* union_ms) {
    if (klass < 0 || klass >= KC_COUNT || !union_ms) FAIL(c, "maus_profile_union_ms: bad arguments");
    prof_resolve(c);
    auto iv = c->prof_iv[klass];
    std::sort(iv.begin(), iv.end());
    double tot = 0.0; float cs = 0.f, ce = -1.f;
    for (auto& p : iv) {
        if (ce < 0.f) { cs = p.first; ce = p.second; }
        else if (p.first <= ce) ce = std::max(ce, p.second);
        else { tot += ce - cs; cs = p.first; ce = p.second; }
    }
    if (ce >= 0.f) tot += ce - cs;
    *union_ms = tot;
    return 0;
}

int maus_profile_read(maus_ctx* c, int klass, int* launches, double* total_ms, double* flops, double* bytes) {
    if (klass < 0 || klass >= KC_COUNT) FAIL(c, "maus_profile_read: bad class");
    prof_resolve(c);
    if (launches) *launches = c->launches[klass];
    if (total_ms) *total_ms = c->ms[klass];
    if (flops) *flops = c->flops[klass];
    if (bytes) *bytes = c->bytes[klass];
    return 0;
}

}  // extern "C"

// accessors for the GMRES translation unit
hipStream_t maus_ctx_stream(maus_ctx* c) { return c->st; }
void maus_ctx_set_error(maus_ctx* c, const char* m) { c->err = m; }
