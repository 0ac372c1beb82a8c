// Internal definitions shared by capi.hip and gmres.hip (not part of the C ABI).
#pragma once
#include "common.h"
#include "../../include/maus_hip.h"

#include <map>
#include <string>
#include <algorithm>
#include <vector>

// ---- kernels / drivers implemented in the other translation units ---------------------
#include "luws.h"
#include "mtplan.h"
void maus_lu_factor(const LuWs& w, int nbo);
void maus_lu_backsolve(const LuWs& w, c128* Wpop, long ldw, const int* d_slots, c128* xout_dense);
void maus_build_h(const LuWs& w, const c128* A, const c128* d_shift, const double* d_psi, int rhs_mode,
                  const c128* X, long ldx, const int* d_slots, const c128* bvec, int pert_mode, const double* d_U, int tiled);
void maus_load_h(const LuWs& w, const c128* d_Ain, const c128* d_bin);
int maus_lu_max_npad();
size_t maus_lu_mw_sync_bytes();
void maus_mt_copy_states(hipStream_t st, uint32_t* states, const uint32_t* base, int count);
#include "mtjump.h"
int maus_mt_tap_split();
void maus_mt_jump(hipStream_t st, uint32_t* states, const int* sel, const int* mult, int nsel, const MausJumpPolys& P, int src_off);
int maus_mt_zero_tap();
void maus_build_h_mt(hipStream_t st, const c128* A, int n, int npad, long ldh, long strideH, c128* H, int G, int S, long E,
                     const c128* d_shift, const double* d_psi, int rhs_mode, const c128* X, long ldx, const int* d_slots,
                     const c128* bvec, const uint32_t* states, const int* extra, const int* rpos, int* flags, int tiled);
int maus_mt_jump_poly(uint64_t J, uint64_t* out312);
void maus_zgemm_launch_idx(hipStream_t st, int M, int N, int K, const c128* A, long lda, long sA,
                           const c128* B, long ldb, long sB, c128* C, long ldc, long sC,
                           double alpha, int beta, int batch, int blay, bool conja, bool conjb,
                           const int* a_rows, const int* c_rows);
void maus_launch_rayleigh_dots(hipStream_t st, const c128* X, const c128* Y, long ld, const int* slots, int count, int n, c128* num, c128* den);
void maus_launch_relax(hipStream_t st, c128* X, const c128* W, long ld, const int* slots, int count, int n, const c128* alpha, int normalise, double* norm_out);
void maus_launch_residual(hipStream_t st, int kind, const c128* X, const c128* Y, long ld, const int* slots, int count, int n,
                          const c128* lam, const c128* bvec, double* resid, int* finite);
void maus_launch_svd_resid(hipStream_t st, const c128* Yv, const c128* Uv, long ld, const int* slots, int count, int n,
                           const c128* sigma, double* out, int accumulate, int* finite);
void maus_launch_norm_scale(hipStream_t st, const c128* S, c128* D, long ld, const int* slots, int count, int n, double* norm_out, int stride_out, int off_out);
void maus_launch_norm(hipStream_t st, const c128* S, long ld, const int* slots, int count, int n, double* norm_out, int stride_out, int off_out);
void maus_launch_herm_pick(hipStream_t st, const c128* S, long lds_, c128* X, long ldx, const int* slots, int count, const c128* V, int n, int* idx_out, double* norm_out);
int maus_gmres_run(maus_ctx* ctx, const int* slots, int count, const double* shift, const double* psi, int rhs_mode,
                   const int32_t* use_jacobi, double rtol, int restart, int maxiter, int32_t* info_out, int32_t* inner_out, int32_t* status,
                   const c128* Hdense, long ldh, long strideH, int32_t* jacobi_out);
int maus_jacobi_check_run(maus_ctx* ctx, int count, const double* shift, const double* psi, int32_t* ok);

// ---- context ---------------------------------------------------------------------------
struct ProfRec { int klass; hipEvent_t e0, e1; double flops, bytes, weight; };

struct maus_ctx {
    int device = 0;
    hipStream_t st = nullptr;
    std::string err;
    c128* A = nullptr; int rows = 0, cols = 0;      // problem matrix
    c128* b = nullptr; int bn = 0;                  // rhs
    c128* V = nullptr; int vn = 0;                  // eigenvectors (Hermitian shortcut)
    c128* hq = nullptr; c128* htau = nullptr; int hqn = 0;   // Householder reflectors of maus_herm_tridiag, until the back-transformation (herm.hip)
    double* hz = nullptr; int hzn = 0;                       // eigenvectors of T (maus_herm_tridiag_eig), until the back-transformation
    int cap = 0; long ldp = 0;                      // population
    c128 *X = nullptr, *U = nullptr, *W = nullptr, *Y = nullptr;
    // per-call scalar staging (device), sized for `scal_cap` candidates
    int scal_cap = 0;
    int *d_slots = nullptr, *d_i1 = nullptr, *d_i2 = nullptr;
    // Y[slot] = A X[slot] is known to hold for the slots stamped with the current epoch (capi.hip: av_*): the SVD residual of one
    // loop body leaves the product that the power step of the next one starts with (AMS:295-298 / 228)
    std::vector<uint32_t> av_stamp;
    uint32_t av_epoch = 1;
    // S[slot] = A^H U[slot], the unscaled second product of the SVD power step (AMS:240), kept for the residual of the same loop
    // body (AMS:298: the same product of the same u); allocated by the first power step; stamps as for Y
    c128* S = nullptr;
    int Scap = 0;
    std::vector<uint32_t> ahu_stamp, prop_stamp;   // prop_stamp: rows of the latest proposal (S holds A^H of ITS u until committed)
    uint32_t ahu_epoch = 1, prop_epoch = 1;
    c128 *d_c1 = nullptr, *d_c2 = nullptr;
    double *d_r1 = nullptr, *d_r2 = nullptr;
    // LU workspace
    c128* H = nullptr; size_t Hbytes = 0; int Hg = 0; int Hnpad = 0; int ws_allocs = 0;
    bool ws_at_limit = false;      // the workspace has reached what this device / MAUS_LU_BATCH allow: never re-allocated again for this npad
    // multi-workgroup panel (lu.hip): off when the device is shared with other processes (maus_set_shared_device) and
    // after its first rendezvous time-out on this context; mw_aborts counts the batches that were repeated without it
    bool shared_device = false; bool mw_disabled = false; int mw_aborts = 0;
    int *ipiv = nullptr, *perm = nullptr, *info = nullptr, *flags = nullptr; void* mw_sync = nullptr;
    int* ident = nullptr;                     // 0, 1, .., npad - 1: the row list of products on the logical-order U array (blocked back substitution)
    double* Upert = nullptr; size_t Ubytes = 0;
    // device-side MT19937 regeneration (mtdev.hip)
    // one buffer set per sub-batch stream: the host prepares sub-batch s+1 while the jump / build kernels of sub-batch s
    // still read theirs
    struct MtBuf { uint32_t* states = nullptr; int* ints = nullptr; uint32_t* base = nullptr; int cap = 0; size_t int_cap = 0; };
    std::vector<MtBuf> mt_bufs;
    struct MtTaps { int* taps; int ntap16, nlo16; };
    std::map<uint64_t, MtTaps> mt_taps;                 // J -> device tap list of x^J mod phi in the two-window form of mt_jump_kernel
    MausMtPlan mt_plan;                                 // host plan + staging image of the current sub-batch (mtplan.cpp)
    // optional sub-batch streams (MAUS_LU_STREAMS > 1): bandwidth-bound phases of one sub-batch beside the
    // MFMA-bound trailing updates of another
    std::vector<hipStream_t> lu_st; std::vector<hipEvent_t> lu_done; hipEvent_t ev_stage = nullptr;
    hipStream_t prof_st = nullptr;
    // history store (SURVEY f-4): rows appended on the device, oldest chunks spilled to host memory beyond a byte budget
    struct HistChunk { c128* dev = nullptr; c128* host = nullptr; long rows = 0; long cap = 0; };
    std::vector<HistChunk> hist; long hist_len = 0; long hist_rows = 0; size_t hist_dev_bytes = 0;
    long hist_gen = 0;             // bumped whenever the store is dropped: indices handed out before are stale
    // population sharding (comm.hip): one RCCL communicator per context, a staging buffer, and the host wall time spent
    // inside collectives (reported per rank by bench.py)
    void* comm = nullptr; int comm_rank = 0, comm_world = 0;
    void* comm_buf = nullptr; size_t comm_buf_bytes = 0;
    long comm_calls = 0; double comm_bytes = 0, comm_ms = 0;
    // generic scratch (host-GEMM / host-LU test entry points, GMRES)
    void* scratch = nullptr; size_t scratch_bytes = 0;
    void* pin = nullptr; size_t pin_bytes = 0;           // pinned host staging buffer (maus_stage_h2d / _d2h, maus_pop_put / _get)
    hipEvent_t pin_ev[2] = {nullptr, nullptr}; bool pin_failed = false;
    char* pin_small = nullptr; size_t pin_small_off = 0;   // pinned ring for small asynchronous uploads (maus_h2d)
    hipEvent_t pin_small_ev[2] = {nullptr, nullptr}; long pin_small_lap = 0; bool pin_small_mid = false;   // one event per half of the ring (maus_h2d)
    // measurement
    hipEvent_t t0 = nullptr, t1 = nullptr;
    bool prof_on = false;
    int prof_mode = 1;            // 1: every launch of every class; 2: every 5th launch of the zgemm classes only
    long prof_seq = 0; bool prof_skip = false;
    long prof_cnt[KC_COUNT] = {0}; int prof_stride_big = 1, prof_stride_small = 0; double prof_weight = 1.0;
    hipEvent_t prof_origin = nullptr;                   // recorded on `st` when profiling is switched on
    std::vector<std::pair<float, float>> prof_iv[KC_COUNT];   // [start, end] ms since prof_origin of every bracketed launch
    int total_launches[KC_COUNT] = {0};
    std::vector<ProfRec> pending;
    std::vector<hipEvent_t> pool;
    hipEvent_t cur0 = nullptr;
    int launches[KC_COUNT] = {0}; double ms[KC_COUNT] = {0}, flops[KC_COUNT] = {0}, bytes[KC_COUNT] = {0};
};

extern thread_local std::string g_err;

#define FAIL(ctx, msg) do { (ctx)->err = (msg); return -1; } while (0)
#define HIPCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
        char buf_[512]; snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
        (ctx)->err = buf_; return -1; } } while (0)


int maus_matrix_reserve(maus_ctx* c, int rows, int cols);     // capi.hip: device room for the problem matrix
int ensure_scalars(maus_ctx* c, int count);
int ensure_scratch(maus_ctx* c, size_t bytes);
int check_slots(maus_ctx* c, const int* slots, int count);
int upload_slots(maus_ctx* c, const int* slots, int count);
// no row of Y is known to hold A X any more (see maus_ctx::av_stamp)
inline void maus_av_drop_all(maus_ctx* c) {
    if (++c->av_epoch == 0) { c->av_epoch = 1; std::fill(c->av_stamp.begin(), c->av_stamp.end(), 0u); }
    if (++c->ahu_epoch == 0) { c->ahu_epoch = 1; std::fill(c->ahu_stamp.begin(), c->ahu_stamp.end(), 0u); }
}
// Host <-> device copies of anything larger than a few KB go through the context's pinned buffer, never straight from / into the
// caller's memory (capi.hip).  Both are synchronous with respect to `st`.
int maus_pin_ready(maus_ctx* c);
int maus_stage_h2d(maus_ctx* c, void* dst_dev, const void* src_host, size_t bytes, hipStream_t st);
int maus_stage_d2h(maus_ctx* c, void* dst_host, const void* src_dev, size_t bytes, hipStream_t st);
int maus_h2d(maus_ctx* c, void* dst_dev, const void* src_host, size_t bytes, hipStream_t st);   // <= 16 KB: asynchronous (the source is copied out before it returns)
int maus_d2h(maus_ctx* c, void* dst_host, const void* src_dev, size_t bytes, hipStream_t st);
void prof_tick(void* ud, int klass, int phase, double flops, double bytes);

struct ProfScope {
    maus_ctx* c; int k; double f, b;
    ProfScope(maus_ctx* c_, int k_, double f_ = 0, double b_ = 0) : c(c_), k(k_), f(f_), b(b_) { prof_tick(c, k, 0, 0, 0); }
    ~ProfScope() { prof_tick(c, k, 1, f, b); }
};
