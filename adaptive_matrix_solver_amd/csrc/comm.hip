// Population sharding over the GPUs of one node: the collectives of the candidate step on RCCL over xGMI
// (include/maus_hip.h "population sharding"; SURVEY §8b maus_comm_init_all / maus_allgather_records, §8e).
//
// The reference is one sequential loop over the candidates (AMS:574-576); nothing of this has a counterpart there.  One
// process per GPU owns one context; within an iteration a candidate's step reads only (A, b, strategy) and its own state,
// so the only exchange is an ALL-GATHER: of the per-candidate records the host bookkeeping consumes (AMS:424-475: residual,
// stuckness, weight, ...) and of the candidate rows the owners updated.  Everything runs on the context's stream, so it is
// ordered with the kernels that produce and consume the rows.
//
// librccl is loaded on first use (dlopen), not linked: the library must load -- and every non-collective entry point
// work -- on a box without RCCL, and a process that never shards never pays for it.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <vector>

#include "ctx.h"
// Types only: every call goes through dlsym.  Without the RCCL development headers the few declarations the calls need are
// spelled out here (they are part of NCCL's stable C ABI), so that the library still builds.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0 } ncclDataType_t;
#endif

namespace {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

Rccl* rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return &r;
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) { r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (r.h) break; }
    if (!r.h) { const char* e = dlerror(); r.err = std::string("librccl not found: ") + (e ? e : "dlopen failed"); return &r; }
#define SYM(field, name) do { *(void**)(&r.field) = dlsym(r.h, name); if (!r.field) { r.err = std::string("librccl lacks ") + name; r.h = nullptr; return &r; } } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllGather, "ncclAllGather");
    SYM(Broadcast, "ncclBroadcast");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    return &r;
}

#define NCCLCHK(ctx, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { \
        char buf_[512]; snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, rccl()->GetErrorString(r_), __FILE__, __LINE__); \
        (ctx)->err = buf_; return -1; } } while (0)

struct Timer {
    maus_ctx* c; std::chrono::steady_clock::time_point t0; double bytes;
    Timer(maus_ctx* c_, double b) : c(c_), t0(std::chrono::steady_clock::now()), bytes(b) {}
    ~Timer() { c->comm_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
               c->comm_calls++; c->comm_bytes += bytes; }
};

int ensure_comm_buf(maus_ctx* c, size_t bytes) {
    if (bytes <= c->comm_buf_bytes) return 0;
    if (c->comm_buf) { HIPCHK(c, hipStreamSynchronize(c->st)); (void)hipFree(c->comm_buf); c->comm_buf = nullptr; c->comm_buf_bytes = 0; }
    size_t want = std::max(bytes, (size_t)1 << 20);
    HIPCHK(c, hipMalloc(&c->comm_buf, want));
    c->comm_buf_bytes = want;
    return 0;
}

c128* pop_array(maus_ctx* c, int which) {
    switch (which) { case MAUS_POP_X: return c->X; case MAUS_POP_U: return c->U; case MAUS_POP_W: return c->W; case MAUS_POP_Y: return c->Y; }
    return nullptr;
}

// send[i][0:len] <- P[slots[i]][0:len]  (rows this rank owns, packed)
__global__ void pack_rows_kernel(const c128* __restrict__ P, long ld, const int* __restrict__ slots, int len, c128* __restrict__ send) {
    const c128* s = P + (long)slots[blockIdx.x] * ld;
    c128* d = send + (long)blockIdx.x * len;
    for (int k = threadIdx.x; k < len; k += blockDim.x) d[k] = s[k];
}

// P[slots[off_r + i]][0:len] <- recv[r][i][0:len] for every rank r != me and i < counts[r]; blockIdx.x walks the flat slot list
__global__ void unpack_rows_kernel(c128* __restrict__ P, long ld, const int* __restrict__ slots, const int* __restrict__ owner,
                                   const int* __restrict__ index_in_rank, int len, long cmax, const c128* __restrict__ recv, int me) {
    const int f = blockIdx.x, r = owner[f];
    if (r == me) return;                                           // own rows are already in place
    const c128* s = recv + ((long)r * cmax + index_in_rank[f]) * len;
    c128* d = P + (long)slots[f] * ld;
    for (int k = threadIdx.x; k < len; k += blockDim.x) d[k] = s[k];
}

}  // namespace

extern "C" {

int maus_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int maus_comm_unique_id(char* id_out) {
    if (!id_out) return -1;
    Rccl* r = rccl();
    if (!r->h) { g_err = r->err; return -1; }
    ncclUniqueId id;
    ncclResult_t rc = r->GetUniqueId(&id);
    if (rc != ncclSuccess) { g_err = std::string("ncclGetUniqueId failed: ") + r->GetErrorString(rc); return -1; }
    static_assert(sizeof(id.internal) == MAUS_COMM_ID_BYTES, "RCCL unique id size");
    memcpy(id_out, id.internal, MAUS_COMM_ID_BYTES);
    return 0;
}

int maus_comm_init(maus_ctx* c, int rank, int world, const char* id_bytes) {
    if (!c || !id_bytes || world < 1 || rank < 0 || rank >= world) { if (c) c->err = "maus_comm_init: bad arguments"; return -1; }
    if (c->comm) FAIL(c, "maus_comm_init: this context already has a communicator");
    Rccl* r = rccl();
    if (!r->h) FAIL(c, r->err.c_str());
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(id.internal, id_bytes, MAUS_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    NCCLCHK(c, r->CommInitRank(&comm, world, id, rank));
    c->comm = (void*)comm; c->comm_rank = rank; c->comm_world = world;
    c->comm_ms = 0; c->comm_calls = 0; c->comm_bytes = 0;
    return 0;
}

int maus_comm_destroy(maus_ctx* c) {
    if (!c || !c->comm) return 0;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->st);
    (void)rccl()->CommDestroy((ncclComm_t)c->comm);
    c->comm = nullptr; c->comm_world = 0; c->comm_rank = 0;
    if (c->comm_buf) { (void)hipFree(c->comm_buf); c->comm_buf = nullptr; c->comm_buf_bytes = 0; }
    return 0;
}

int maus_comm_info(maus_ctx* c, int* rank_out, int* world_out) {
    if (!c) return -1;
    if (rank_out) *rank_out = c->comm ? c->comm_rank : 0;
    if (world_out) *world_out = c->comm ? c->comm_world : 0;
    return 0;
}

// recv[r][0:bytes_per_rank] <- rank r's send[0:bytes_per_rank], host buffers (the scalar records the host bookkeeping reads:
// tens of bytes per candidate, latency-bound -- staged through the context's device buffer, one ncclAllGather).
int maus_comm_allgather_records(maus_ctx* c, const void* send, size_t bytes_per_rank, void* recv) {
    if (!c->comm) FAIL(c, "maus_comm_allgather_records: no communicator (maus_comm_init)");
    if (bytes_per_rank == 0) return 0;
    if (!send || !recv) FAIL(c, "maus_comm_allgather_records: null buffer");
    const size_t W = (size_t)c->comm_world, chunk = (bytes_per_rank + 15) / 16 * 16;
    Timer tm(c, (double)bytes_per_rank * W);
    if (ensure_comm_buf(c, chunk * (W + 1))) return -1;
    char* d_send = (char*)c->comm_buf; char* d_recv = d_send + chunk;
    // host buffers cross through the context's pinned buffer (capi.hip, "pinned staging"): the caller's arrays are short-lived
    if (maus_stage_h2d(c, d_send, send, bytes_per_rank, c->st)) return -1;
    NCCLCHK(c, rccl()->AllGather(d_send, d_recv, chunk, ncclChar, (ncclComm_t)c->comm, c->st));
    if (chunk == bytes_per_rank) return maus_stage_d2h(c, recv, d_recv, chunk * W, c->st);
    for (size_t r = 0; r < W; ++r)
        if (maus_stage_d2h(c, (char*)recv + r * bytes_per_rank, d_recv + r * chunk, bytes_per_rank, c->st)) return -1;
    return 0;
}

// After a sharded step every rank holds fresh rows only for the candidates it stepped.  slots = the slots of all candidates
// of the step grouped by owner rank (rank 0's first; counts[r] of them for rank r; the same list on every rank).  The
// owners' rows are packed, all-gathered device to device and scattered into the population array `which` of every rank:
// one pack kernel, one ncclAllGather over xGMI, one unpack kernel -- 16 MB per step at n = 4096 / 256 candidates.
int maus_comm_allgather_rows(maus_ctx* c, int which, const int* slots, const int* counts, int len) {
    maus_av_drop_all(c);                                 // rows of X change under the products kept in Y (capi.hip: av_*)
    if (!c->comm) FAIL(c, "maus_comm_allgather_rows: no communicator (maus_comm_init)");
    c128* P = pop_array(c, which);
    if (!P) FAIL(c, "maus_comm_allgather_rows: population not reserved / bad array id");
    if (!counts || len <= 0 || len > c->ldp) FAIL(c, "maus_comm_allgather_rows: bad arguments");
    const int W = c->comm_world, me = c->comm_rank;
    long total = 0, cmax = 0;
    for (int r = 0; r < W; ++r) { if (counts[r] < 0) FAIL(c, "maus_comm_allgather_rows: negative count"); total += counts[r]; cmax = std::max<long>(cmax, counts[r]); }
    if (total == 0) return 0;
    if (!slots) FAIL(c, "maus_comm_allgather_rows: null slot list");
    if (check_slots(c, slots, (int)total)) return -1;
    const size_t row_bytes = sizeof(c128) * (size_t)len;
    Timer tm(c, (double)row_bytes * total);
    // device staging: [flat slots | owner | index in rank] ints, then send (cmax rows) and recv (W * cmax rows)
    const size_t ints_bytes = ((sizeof(int) * 3 * (size_t)total) + 255) / 256 * 256;
    if (ensure_comm_buf(c, ints_bytes + row_bytes * (size_t)cmax * (W + 1))) return -1;
    std::vector<int> h(3 * (size_t)total);
    long off = 0, my_off = 0;
    for (int r = 0; r < W; ++r) {
        if (r == me) my_off = off;
        for (int i = 0; i < counts[r]; ++i) { h[off + i] = slots[off + i]; h[total + off + i] = r; h[2 * total + off + i] = i; }
        off += counts[r];
    }
    int* d_ints = (int*)c->comm_buf;
    c128* d_send = (c128*)((char*)c->comm_buf + ints_bytes);
    c128* d_recv = d_send + (size_t)cmax * len;
    if (maus_h2d(c, d_ints, h.data(), sizeof(int) * h.size(), c->st)) return -1;
    HIPCHK(c, hipStreamSynchronize(c->st));                         // h is about to go out of scope
    if (counts[me] > 0)
        hipLaunchKernelGGL(pack_rows_kernel, dim3(counts[me]), dim3(256), 0, c->st, P, c->ldp, d_ints + my_off, len, d_send);
    NCCLCHK(c, rccl()->AllGather(d_send, d_recv, row_bytes * (size_t)cmax, ncclChar, (ncclComm_t)c->comm, c->st));
    hipLaunchKernelGGL(unpack_rows_kernel, dim3((unsigned)total), dim3(256), 0, c->st, P, c->ldp, d_ints, d_ints + total, d_ints + 2 * total,
                       len, cmax, d_recv, me);
    HIPCHK(c, hipStreamSynchronize(c->st));
    HIPCHK(c, hipGetLastError());
    return 0;
}

// Broadcast of a host buffer from `root` (start-up data that only rank 0 computes: matrix diagnostics, the eigenvalues of
// the Hermitian shortcut), staged through the device in chunks.
int maus_comm_bcast(maus_ctx* c, void* buf, size_t bytes, int root) {
    if (!c->comm) FAIL(c, "maus_comm_bcast: no communicator (maus_comm_init)");
    if (bytes == 0) return 0;
    if (!buf || root < 0 || root >= c->comm_world) FAIL(c, "maus_comm_bcast: bad arguments");
    Timer tm(c, (double)bytes);
    const size_t chunk = (size_t)64 << 20;
    if (ensure_comm_buf(c, std::min(bytes, chunk))) return -1;
    for (size_t off = 0; off < bytes; off += chunk) {
        const size_t nb = std::min(chunk, bytes - off);
        if (c->comm_rank == root && maus_stage_h2d(c, c->comm_buf, (char*)buf + off, nb, c->st)) return -1;
        NCCLCHK(c, rccl()->Broadcast(c->comm_buf, c->comm_buf, nb, ncclChar, root, (ncclComm_t)c->comm, c->st));
        if (c->comm_rank != root) { if (maus_stage_d2h(c, (char*)buf + off, c->comm_buf, nb, c->st)) return -1; }
        else HIPCHK(c, hipStreamSynchronize(c->st));
    }
    return 0;
}

// The eigenvector matrix of the Hermitian shortcut (AMS:161, once per matrix here): uploaded by `root` with
// maus_set_eigvecs, broadcast device to device (1 GiB at n = 8192) -- the other ranks never decompose the matrix.
int maus_comm_bcast_eigvecs(maus_ctx* c, int n, int root) {
    if (!c->comm) FAIL(c, "maus_comm_bcast_eigvecs: no communicator (maus_comm_init)");
    if (n <= 0 || n != c->rows || n != c->cols || root < 0 || root >= c->comm_world) FAIL(c, "maus_comm_bcast_eigvecs: bad arguments");
    const size_t bytes = sizeof(c128) * (size_t)n * n;
    Timer tm(c, (double)bytes);
    HIPCHK(c, hipStreamSynchronize(c->st));
    if (c->comm_rank == root) { if (!c->V || c->vn != n) FAIL(c, "maus_comm_bcast_eigvecs: root has no eigenvectors (maus_set_eigvecs)"); }
    else if (n != c->vn) { if (c->V) (void)hipFree(c->V); c->V = nullptr; c->vn = 0; HIPCHK(c, hipMalloc((void**)&c->V, bytes)); c->vn = n; }
    NCCLCHK(c, rccl()->Broadcast(c->V, c->V, bytes, ncclChar, root, (ncclComm_t)c->comm, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    return 0;
}

// The problem matrix of a sharded run (A replicated on every GPU): `root` alone uploads it from the host -- one staged
// copy through one pinned buffer instead of one per rank, 8 x 1 GiB at n = 8192 -- and the others receive it device to device.
// The root's outcome travels first (one word): an upload that fails there fails this call on every rank instead of leaving
// the others inside a broadcast the root never enters.  `a` is read on the root only; NULL there = the matrix of that shape is
// already resident on the root's device (its start-up diagnostics put it there) and is broadcast as it is.
int maus_comm_set_matrix(maus_ctx* c, const double* a, int rows, int cols, int root) {
    if (!c->comm) FAIL(c, "maus_comm_set_matrix: no communicator (maus_comm_init)");
    if (rows <= 0 || cols <= 0 || root < 0 || root >= c->comm_world) FAIL(c, "maus_comm_set_matrix: bad arguments");
    const size_t bytes = sizeof(c128) * (size_t)rows * cols;
    Timer tm(c, (double)bytes);
    int rc;
    if (c->comm_rank != root) rc = maus_matrix_reserve(c, rows, cols);
    else if (a) rc = maus_set_matrix(c, a, rows, cols);
    else { rc = (c->A && c->rows == rows && c->cols == cols) ? 0 : -1; if (rc) c->err = "no resident matrix of that shape on the root"; }
    std::string local_err = (rc != 0) ? c->err : std::string();
    if (ensure_comm_buf(c, 64)) return -1;
    int* d_status = (int*)c->comm_buf;
    int h_status = (c->comm_rank == root) ? rc : 0;
    HIPCHK(c, hipMemcpyAsync(d_status, &h_status, sizeof(int), hipMemcpyHostToDevice, c->st));
    NCCLCHK(c, rccl()->Broadcast(d_status, d_status, sizeof(int), ncclChar, root, (ncclComm_t)c->comm, c->st));
    HIPCHK(c, hipMemcpyAsync(&h_status, d_status, sizeof(int), hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    if (h_status != 0) {
        if (c->comm_rank == root) FAIL(c, "maus_comm_set_matrix: upload failed on the root: " + local_err);
        FAIL(c, "maus_comm_set_matrix: the root's upload of the matrix failed");
    }
    if (rc != 0) FAIL(c, "maus_comm_set_matrix: no device memory for the matrix on this rank: " + local_err);   // (the peers' broadcast below then fails too: RCCL's own error)
    NCCLCHK(c, rccl()->Broadcast(c->A, c->A, bytes, ncclChar, root, (ncclComm_t)c->comm, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    return 0;
}

int maus_comm_stats(maus_ctx* c, long* calls_out, double* bytes_out, double* ms_out, int reset) {
    if (!c) return -1;
    if (calls_out) *calls_out = c->comm_calls;
    if (bytes_out) *bytes_out = c->comm_bytes;
    if (ms_out) *ms_out = c->comm_ms;
    if (reset) { c->comm_calls = 0; c->comm_bytes = 0; c->comm_ms = 0; }
    return 0;
}

}  // extern "C"
