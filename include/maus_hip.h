/* libmaus_hip -- C ABI of the MI355X (gfx950) hot path of MAUS.
 *
 * The reference (Kier73/Adaptive-Matrix-Solver, `Adaptive_Matrix_Solver_0.1.py`,
 * cited as AMS:line) is pure Python with no FFI layer; the boundary below is what
 * a ctypes binding of its per-candidate inner loop binds.
 * Each entry point names the reference lines it replaces.
 *
 * Conventions
 *  - Plain pointers and sizes only.  Complex data is interleaved (re, im) double,
 *    i.e. NumPy complex128, C-contiguous (row-major).
 *  - Every function returns 0 on success, <0 on an API/HIP error
 *    (maus_last_error() gives the text).  Nothing throws across the boundary.
 *  - NUMERICAL failures are not errors: they come back per candidate in
 *    `status[]` (0 ok; k>0 exact zero pivot at column k, LAPACK `info`
 *    convention, AMS:59 -> LinAlgError; -1 non-finite input matrix/rhs, -2
 *    non-finite result, AMS:94-95 -> ValueError), so the Python retry ladder
 *    (AMS:98-104) is reproduced exactly.
 *  - Host buffers are caller-owned and are only read / written by plain memcpy
 *    while a call is in progress: every transfer goes through a pinned buffer
 *    of the context (a caller array handed to the runtime would stay
 *    registered with the driver and stall the GPU queues when the caller frees
 *    it).  Device buffers are owned by the context.
 *  - One context per GPU, not thread-safe; calls are synchronous as seen by the
 *    caller (work is enqueued on the context's own HIP stream and joined before
 *    results are returned).
 *  - A "slot" is a row of the device-resident population arrays (one candidate
 *    vector per contiguous row).
 */
#ifndef MAUS_HIP_H
#define MAUS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct maus_ctx maus_ctx;

/* population arrays (`which`) */
enum { MAUS_POP_X = 0,   /* v_k (eig) / x_k (linear) / right_v_k (SVD)  AMS:120-121 */
       MAUS_POP_U = 1,   /* u_k (SVD)                                    AMS:121 */
       MAUS_POP_W = 2,   /* raw solver result new_vec_raw                AMS:278 */
       MAUS_POP_Y = 3 }; /* scratch: last A@X product */

/* residual kinds (AMS:295-301) */
enum { MAUS_EIG = 1, MAUS_LINEAR = 2, MAUS_SVD = 3 };

/* perturbation modes of the dense regulariser (AMS:49-50) */
enum { MAUS_PERT_NONE = 0,      /* reg = psi*I only (the 0.15*psi random term is dropped; the caller still
                                   advances the NumPy stream)                                              */
       MAUS_PERT_UNIFORM = 1,   /* caller supplies the two rand(N,N) draws; device forms
                                   ((U1-.5)+i(U2-.5))*psi*0.15 with the reference's rounding order       */
       MAUS_PERT_MT19937 = 2 }; /* device regenerates the legacy NumPy MT19937 stream from a supplied
                                   624-word state per candidate (bit-identical draws)                    */

/* MAUS_PERT_MT19937 descriptor: legacy NumPy RandomState (np.random.get_state()) + consumption pattern.
 * words_per_candidate and lead_words must be multiples of 2*n*n (one rand(N,N) draw). */
typedef struct {
    uint32_t key[624];
    int32_t pos;                    /* 0..624, as returned by get_state() */
    int32_t reserved;
    uint64_t words_per_candidate;   /* 4*n*n per dense attempt (8*n*n when a swallowed GMRES attempt draws first) */
    uint64_t lead_words;            /* words each candidate skips before the draws it uses */
    const int32_t* ordinals;        /* [count] position of each candidate in the run */
} maus_mt_desc;

/* ---- lifecycle ---------------------------------------------------------- */
int maus_ctx_create(int device, maus_ctx** out);
int maus_ctx_destroy(maus_ctx* ctx);
const char* maus_last_error(const maus_ctx* ctx);
/* name_len bytes of device name, CU count, total and free HBM bytes */
int maus_device_info(maus_ctx* ctx, char* name, int name_len, int* cus, size_t* hbm_total, size_t* hbm_free);
/* library ABI version (compile-time constant) */
int maus_abi_version(void);

/* ---- problem data (AMS:343, 346: everything is complex128) -------------- */
/* Upload the rows x cols problem matrix.  Replaces passing `current_matrix_A`
 * by reference into every step (AMS:576). */
int maus_set_matrix(maus_ctx* ctx, const double* a_c128, int rows, int cols);
/* Upload b (AMS:146, 275). */
int maus_set_rhs(maus_ctx* ctx, const double* b_c128, int n);

/* ---- device-resident population ----------------------------------------- */
int maus_pop_reserve(maus_ctx* ctx, int capacity);
int maus_pop_capacity(maus_ctx* ctx);
/* host (count x len, C-contiguous complex128) -> slots[i] of array `which`; len = vector length */
int maus_pop_put(maus_ctx* ctx, int which, const int* slots, int count, const double* host_c128, int len);
int maus_pop_get(maus_ctx* ctx, int which, const int* slots, int count, double* host_c128, int len);

/* Device address of population array `which` (capacity x ld complex128 elements, row-major) after joining the context's
 * stream -- for a caller that exchanges candidate rows between GPUs with its own collective library (RCCL through
 * torch.distributed in dist.py) without bouncing them through the host.  The pointer is invalidated by
 * maus_pop_reserve / maus_set_matrix with a different vector length. */
int maus_pop_device_ptr(maus_ctx* ctx, int which, void** ptr_out, long* ld_out, int* capacity_out);

/* Device-side copy of the rows `slots` from population array which_src to which_dst (no host traffic).  Used to
 * snapshot the candidates of a speculative batch before the relaxed update overwrites them, so that a run can be
 * restarted behind an RNG event (E4 tiny-norm re-initialisation, AMS:283) exactly as the sequential reference
 * would continue. */
int maus_pop_copy(maus_ctx* ctx, int which_dst, int which_src, const int* slots, int count);

/* ---- device-backed history (AMS:126, 303-304: param_history keeps every iterate of every candidate) ---- */
/* Append the first `len` entries of rows `slots` of population array `which` to the context's history store
 * (device-to-device, no host traffic); the rows get consecutive indices starting at *first_index_out.  Beyond
 * MAUS_HIST_DEVICE_BYTES (default 8 GiB) the oldest rows are spilled to host memory.  maus_hist_get fetches rows by
 * index (from wherever they live) into host_c128[count][len]; maus_hist_clear drops everything. */
int maus_hist_append(maus_ctx* ctx, int which, const int* slots, int count, int len, int64_t* first_index_out);
int maus_hist_get(maus_ctx* ctx, const int64_t* indices, int count, int len, double* host_c128);
int maus_hist_clear(maus_ctx* ctx);
/* Generation of the history store: incremented whenever the store is dropped (maus_hist_clear, maus_set_matrix with a
 * different vector length).  A caller that keeps row indices (solver._HistRef) records the generation with them and
 * refuses to resolve an index of an older generation instead of reading somebody else's rows. */
int64_t maus_hist_generation(maus_ctx* ctx);

/* ---- phases of update_solution_step, batched over `count` candidates ---- */
/* Y[slot] = A @ X[slot]; num = vdot(v, A@v), den = vdot(v, v)      AMS:264-268.
 * num_c128: count complex; den_c128: count complex (imag is the rounded sum, ~0). */
int maus_matvec_rayleigh(maus_ctx* ctx, const int* slots, int count, double* num_c128, double* den_c128);

/* Psi-regularised direct solve, one LU per candidate            AMS:44-59 (+270):
 *   H_k = (A - shift_k I) + (psi_k I + pert_k);   solve H_k w = rhs;   W[slot] <- w
 * rhs_mode 0: rhs = X[slot] (eig, AMS:271); 1: rhs = b (linear, AMS:275).
 * shift_c128[count] complex (0 for linear), psi[count] real.
 * pert: see MAUS_PERT_*; `pert_data` is
 *   UNIFORM: host double[count][2][n][n]   (U1 then U2 per candidate)
 *   MT19937: host maus_mt_desc (below): the NumPy state at the start of the run and, per candidate,
 *            its ordinal in the run; candidate i uses the 4*n*n words starting at
 *            lead_words + ordinals[i] * words_per_candidate
 * status[count]: see conventions. */
int maus_shifted_lu_solve(maus_ctx* ctx, const int* slots, int count, const double* shift_c128,
                          const double* psi, int rhs_mode, int pert_mode, const void* pert_data,
                          int32_t* status);

/* Size the LU workspace ONCE for up to `count` simultaneous n x n systems (the reference's sla.solve allocates its
 * copy of H per call, AMS:59; here H_k of a whole batch are device-resident, 270 MB each at n = 4096, and re-allocating
 * ~100 GB costs seconds).  Capped by 80 % of the free HBM and MAUS_LU_BATCH (default 512); larger batches run in
 * chunks.  The workspace never shrinks; a batch beyond it makes it grow once, to the limit.  capacity_out (may be NULL):
 * matrices the workspace holds after the call. */
int maus_lu_reserve(maus_ctx* ctx, int n, int count, int* capacity_out);

/* Number of times the LU workspace has been (re-)allocated on this context (measurement: a timed region must not
 * contain one). */
int maus_lu_workspace_allocs(maus_ctx* ctx);

/* Robustness of the small-batch panel (no reference counterpart; the reference contract it protects is that every failed
 * solve surfaces to the retry ladder, AMS:94-104).  For batches that run on one stream the base panel of the LU may
 * spread over several workgroups per matrix that rendezvous once per pivot column; that needs all of them resident
 * at once, which holds only while this context has the device to itself.
 *  - maus_set_shared_device(ctx, 1): the caller knows the device is shared with other processes (several ranks of a
 *    `gloo` rehearsal on one GPU): the multi-workgroup panel is never used.
 *  - A rendezvous that times out marks the matrix (LAPACK-style info = INT_MIN).  The library then repeats the whole
 *    batch with one workgroup per matrix, switches the multi-workgroup panel off for the rest of the context's life and
 *    counts the event (maus_lu_mw_aborts).  If the repeat fails too the call returns -1 (maus_last_error: "LU panel
 *    rendezvous timed out"): a half-factored matrix is never reported as status 0. */
int maus_set_shared_device(maus_ctx* ctx, int shared);
int maus_lu_mw_aborts(maus_ctx* ctx);

/* X[slot] <- (1-alpha) X[slot] + alpha W[slot]; norm_out = ||X||_2; if normalise and
 * norm > 1e-10: X *= 1/norm                                        AMS:280-285.
 * alpha_c128[count] complex.  Slots whose norm test fails are left un-normalised (the
 * host re-initialises them, AMS:283). */
int maus_relax_normalise(maus_ctx* ctx, const int* slots, int count, const double* alpha_c128,
                         int normalise, double* norm_out);

/* residual_k and the finite check                                  AMS:295-301, 319-327.
 * kind EIG: ||A v - lam v||; LINEAR: ||A x - b||; SVD: ||A v - s u|| + ||A^H u - s v||
 * (lam_c128[count]: lambda or sigma+0i).  finite_out[i]=1 iff every entry of the
 * candidate's vectors is finite. */
int maus_residual(maus_ctx* ctx, int kind, const int* slots, int count, const double* lam_c128,
                  double* resid_out, int32_t* finite_out);

/* SVD alternating power step                                       AMS:233-242:
 * t = A v; sigma1 = ||t||; u = t / (sigma1 > 1e-10 ? sigma1 : 1); s = A^H u;
 * sigma2 = ||s||; v = s / (sigma2 > 1e-10 ? sigma2 : 1).
 * Outputs per candidate: norms[4] = {||v_in||, sigma1, ||u||, sigma2}. */
int maus_svd_power_step(maus_ctx* ctx, const int* slots, int count, double* norms_out);
/* The same step in two halves, for callers that run it speculatively over a whole population (the reference steps the
 * candidates one after the other, AMS:574-576, and a collapse -- AMS:229-232, 236-239 -- draws random numbers that the
 * candidates behind it must not have passed): propose computes the norms and leaves the candidates' vectors alone (the
 * proposed u in population array 3, the proposed v in array 2), commit makes the proposal the state of the listed
 * candidates.  maus_svd_power_step = propose + commit of the same list. */
int maus_svd_power_propose(maus_ctx* ctx, const int* slots, int count, double* norms_out);
int maus_svd_commit(maus_ctx* ctx, const int* slots, int count);

/* Hermitian shortcut                                               AMS:165-175:
 * given the eigenvector matrix V (n x n, columns = eigenvectors, uploaded once
 * per matrix version with maus_set_eigvecs) pick argmax_j |v^H V[:,j]| per
 * candidate, copy that column into X[slot], normalise.  idx_out[count]. */
int maus_set_eigvecs(maus_ctx* ctx, const double* v_c128, int n);
int maus_herm_match(maus_ctx* ctx, const int* slots, int count, int32_t* idx_out, double* norm_out);

/* The Hermitian eigendecomposition of AMS:161 (scipy.linalg.eigh = LAPACK zheevr: zhetrd -> dstemr -> zunmtr) with the
 * reduction and the back-transformation on the device, once per matrix:
 *   maus_herm_tridiag        A = Q T Q^H of the context's (Hermitian) matrix, zhetrd('L') semantics and reflector
 *                            conventions; d_out[n] / e_out[n-1]: diagonal / subdiagonal of the real T.  The reflectors stay
 *                            on the device until the back-transformation.
 *   maus_herm_tridiag_eig    eigenpairs (lambda, Z) of T on the device: bisection on the Sturm count, eigenvectors from the
 *                            twisted factorisation of T - lambda I (the getvec step of dstemr), no reorthogonalisation.
 *                            w_out[n] ascending; Z stays on the device for maus_herm_backtransform(ctx, NULL, 0).
 *                            diag_out[3] = {smallest eigenvalue gap / ||T||, largest residual component / ||T||, ||T||}: the
 *                            caller keeps the result only for well separated spectra with rounding-level residuals and
 *                            otherwise solves T on the host (scipy.linalg.eigh_tridiagonal = dstemr, the kernel zheevr uses)
 *                            and passes its Z.
 *   maus_herm_backtransform  V = Q Z (zunmtr semantics) from the real eigenvectors of T -- NULL (the device-resident Z of
 *                            maus_herm_tridiag_eig) or z_real[n][n] row-major (column k = k-th
 *                            eigenvector), or with col_major != 0 column-major as LAPACK returns them -- into the context's
 *                            eigenvector matrix, as if set by maus_set_eigvecs.
 * The first row of V is real (Q e_1 = e_1), LAPACK's phase convention. */
int maus_herm_tridiag(maus_ctx* ctx, double* d_out, double* e_out);
/* Frees the reflector store / eigenvectors of T that maus_herm_tridiag(_eig) keep for maus_herm_backtransform: for callers that
 * only wanted the tridiagonal matrix or its eigenvalues (AMS:559, 567 reporting prologue; singular values of the start-up diagnostics). */
int maus_herm_release(maus_ctx* ctx);
/* the context's eigenvector matrix back on the host (v_out[n][n] complex128, row-major; tests, users of evolve()'s report) */
int maus_get_eigvecs(maus_ctx* ctx, double* v_c128_out, int n);
int maus_herm_backtransform(maus_ctx* ctx, const double* z_real, int col_major);
int maus_herm_tridiag_eig(maus_ctx* ctx, const double* d, const double* e, int n, double* w_out, double* diag_out);
/* the eigenvalues of T alone (bisection only, no n x n work arrays): singular values through the Hermitian embedding, spectra of
 * the reporting prologue (AMS:559 / 567) */
int maus_herm_tridiag_eigvals(maus_ctx* ctx, const double* d, const double* e, int n, double* w_out);

/* Gram block of candidate vectors for the distinctness / redundancy tests      AMS:432-437, 443-451, 509-520:
 * out[i*count + j] = vdot(x_i, x_j) = sum_k conj(x_i[k]) x_j[k] over the first `len` entries of rows `slots`
 * of population array `which` (MAUS_POP_X / MAUS_POP_U).  Replaces the reference's pairwise np.vdot calls
 * between converged candidates; the host keeps the reference's greedy order and thresholds. */
int maus_gram(maus_ctx* ctx, int which, const int* slots, int count, int len, double* out_c128);

/* Batched restarted GMRES with optional Jacobi preconditioner       AMS:60-90 ->
 * scipy/sparse/linalg/_isolve/iterative.py:692-841 (restart 20, MGS, Givens, ptol).
 *   H_k = A - shift_k I + psi_k I (the random term of AMS:49-50 is left out: see maus_gmres_pert); x0 = rhs; W[slot] <- x
 * use_jacobi[count]: 1 -> M = diag(1/diag H_k) (caller applies AMS:65/72 gating via
 * maus_jacobi_check below).
 * info_out: 0 converged, maxiter otherwise (SciPy convention); inner_out: inner iterations. */
int maus_gmres(maus_ctx* ctx, const int* slots, int count, const double* shift_c128, const double* psi,
               int rhs_mode, const int32_t* use_jacobi, double rtol, int restart, int maxiter,
               int32_t* info_out, int32_t* inner_out, int32_t* status);
/* The same solver against the reference's FULL H_solve of the GMRES branch (AMS:49-52, 89):
 *   H_k = A - shift_k I + psi_k I + 0.15 psi_k ((U1-.5) + i(U2-.5)),
 * materialised per candidate in the LU workspace by the build kernels of maus_shifted_lu_solve (pert_mode / pert_data
 * as there), matvec = one GEMV per candidate against its own H_k.  For escalated psi (retry ladder, large aggression /
 * stuck factors) where the random term is no longer below the rounding of a matvec; maus_gmres is the fast path below
 * that.  want_jacobi[count]: 1 -> use M = diag(1/diag H_k) provided the AMS:67-72 gate holds on diag(H_k) (evaluated on
 * the device; jacobi_out[count], may be NULL, reports whether it was used).  status -1: non-finite H_k or rhs. */
int maus_gmres_pert(maus_ctx* ctx, const int* slots, int count, const double* shift_c128, const double* psi,
                    int rhs_mode, const int32_t* want_jacobi, int pert_mode, const void* pert_data,
                    double rtol, int restart, int maxiter,
                    int32_t* info_out, int32_t* inner_out, int32_t* status, int32_t* jacobi_out);
/* AMS:67-72 gate: ok[i]=1 iff all 1/diag(H_k) finite and all |diag(H_k)| > 1e-12 */
int maus_jacobi_check(maus_ctx* ctx, int count, const double* shift_c128, const double* psi, int32_t* ok);

/* ---- population sharding over the GPUs of one node: RCCL over xGMI ------------------------------------------------
 * No reference counterpart: AMS:574-576 steps the candidates in one sequential loop.  Within an iteration a candidate's
 * step reads only (A, b, strategy) and its own state (AMS:576), so one process per GPU steps a contiguous block of the
 * active candidates and the only exchange is an all-gather -- of the per-candidate scalar records the host bookkeeping
 * consumes (AMS:424-475, 504-549: residual, stuckness, weight, status ...) and of the rows the owners updated.  librccl is
 * loaded on first use (dlopen); a process that never shards never needs it.  All calls are collective: every rank of
 * the communicator makes the same call with the same sizes.
 *   maus_device_count        HIP devices visible to this process (0 without a GPU; never fails).
 *   maus_comm_unique_id      ONE rank creates the 128-byte id (ncclGetUniqueId) and hands it to the others by any means
 *                            (dist.py: a socket on MASTER_ADDR).
 *   maus_comm_init           ncclCommInitRank on the context's device; one communicator per context.
 *   maus_comm_allgather_records   host buffers: recv[r] <- rank r's send (bytes_per_rank each), rank order.
 *   maus_comm_allgather_rows      device to device on population array `which`: slots[] lists the slots of every stepped
 *                            candidate grouped by owner rank (counts[r] per rank, the same list on every rank); on return
 *                            every rank holds every owner's rows.
 *   maus_comm_bcast          host buffer from `root` to all (start-up diagnostics, eigenvalues: computed by rank 0 only).
 *   maus_comm_bcast_eigvecs  the eigenvector matrix of the Hermitian shortcut (AMS:161; maus_set_eigvecs on `root` only),
 *                            device to device.
 *   maus_comm_set_matrix     maus_set_matrix for a sharded run (A replicated): `root` uploads `a` from the host, the
 *                            other ranks (a may be NULL there) receive it device to device; the root's failure is
 *                            everybody's.  Replaces N staged host copies of AMS:343's matrix by one.
 *   maus_comm_stats          collectives issued, payload bytes, host wall ms inside them (reset != 0 zeroes them). */
#define MAUS_COMM_ID_BYTES 128
int maus_device_count(void);
int maus_comm_unique_id(char* id_out /* MAUS_COMM_ID_BYTES */);
int maus_comm_init(maus_ctx* ctx, int rank, int world, const char* id /* MAUS_COMM_ID_BYTES */);
int maus_comm_destroy(maus_ctx* ctx);
int maus_comm_info(maus_ctx* ctx, int* rank_out, int* world_out);
int maus_comm_allgather_records(maus_ctx* ctx, const void* send, size_t bytes_per_rank, void* recv);
int maus_comm_allgather_rows(maus_ctx* ctx, int which, const int* slots, const int* counts, int len);
int maus_comm_bcast(maus_ctx* ctx, void* host_buf, size_t bytes, int root);
int maus_comm_bcast_eigvecs(maus_ctx* ctx, int n, int root);
int maus_comm_set_matrix(maus_ctx* ctx, const double* a, int rows, int cols, int root);
int maus_comm_stats(maus_ctx* ctx, long* calls_out, double* bytes_out, double* ms_out, int reset);

/* ---- plain batched GEMM on the context's stream (tests, Gram blocks) ----- */
/* C[M,N] = alpha * opA(A)[M,K] * opB(B) + beta * C, host in/out, row-major complex128.
 * b_layout 0: B is K x N; 1: B is N x K (dot-product form).  conj flags apply to A / B. */
int maus_zgemm_host(maus_ctx* ctx, int M, int N, int K, const double* A, const double* B, double* C,
                    int b_layout, int conj_a, int conj_b, double alpha, int beta);
/* LU factor + solve of `count` dense n x n systems given on the host (tests):
 * a_c128[count][n][n], b_c128[count][n] -> x_c128[count][n], status[count]. */
int maus_lu_solve_host(maus_ctx* ctx, int count, int n, const double* a_c128, const double* b_c128,
                       double* x_c128, int32_t* status, int32_t* ipiv_out /* count*n or NULL */);

/* ---- measurement --------------------------------------------------------- */
/* HIP-event timing on the context's stream. */
int maus_timer_start(maus_ctx* ctx);
int maus_timer_stop(maus_ctx* ctx, float* ms_out);
/* Per-kernel-class accounting (event pairs around each launch of the class while enabled).
 * classes: 0 zgemm (LU trailing update with K>=256 / A@X), 1 lu_panel, 2 trsm, 3 (unused since round 2: row-swap sweeps), 4 build_H, 5 backsolve,
 * 6 vector ops, 7..10 zgemm inside the LU recursion with K = 128 / 64 / 32 / 16 */
/* on = 1: event pairs around every launch of every class; on = 2: around the K>=256 zgemm launches only (class 0;
 * long kernels, so cheap enough for a timed region -- full bracketing costs 3-5 % of throughput; MAUS_PROF_STRIDE
 * can thin them out, each sample then stands for `stride` launches); 0: off */
int maus_profile_enable(maus_ctx* ctx, int on);
int maus_profile_read(maus_ctx* ctx, int klass, int* launches, double* total_ms, double* flops, double* bytes);
/* ms during which at least one bracketed launch of the class was executing (union of their intervals over all
 * streams; equals total_ms when nothing overlaps) */
int maus_profile_union_ms(maus_ctx* ctx, int klass, double* union_ms);
int maus_sync(maus_ctx* ctx);

/* ---- legacy NumPy MT19937 stream helpers (host side) ----- */
/* Advance a 624-word MT19937 key + position by `nwords` 32-bit outputs without
 * materialising them (GF(2) jump polynomial; cached per nwords). */
int maus_mt19937_jump(uint32_t* key624, int32_t* pos, uint64_t nwords);

#ifdef __cplusplus
}
#endif
#endif /* MAUS_HIP_H */
