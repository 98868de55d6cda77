/*
 * proxgrad_hip.h -- C ABI of libproxgrad_hip.so: an MI355X (gfx950) native engine for the
 * ForwardBackward / FastForwardBackward inner iteration of ProximalAlgorithms.jl.
 *
 * This header is the drop-in boundary.  Each entry point cites the
 * reference interface it replaces (paths relative to the ProximalAlgorithms.jl checkout).
 * The reference host language is Julia; `ccall` binds these symbols directly (see
 * INTEGRATION.md for the Julia-side glue).  A Python/ctypes host with the same operator
 * and iterator surface lives in `proximalalgorithms.jl_amd/`.
 *
 * This file holds the entry points of SURVEY.md section 8 (the hot path and its "next" rows).  Exports that serve
 * algorithms outside that scope (Davis-Yin's single sweep, the Broyden rank-one update, stream capture / graphs, the
 * L-BFGS image slab) are declared in proxgrad_hip_ext.h.
 *
 * Conventions
 *  - plain C, `extern "C"`, opaque handles, plain pointers and sizes; no C++/torch types.
 *  - every function returns a pg_status (0 = ok, negative = error); pg_last_error() gives the
 *    message of the last failure on the calling thread.
 *  - `dtype`: PG_F32 (Float32) or PG_F64 (Float64) -- the reference's `real(eltype(x0))`.
 *  - vectors are raw DEVICE pointers to `n` (or `m`) contiguous elements of `dtype`, owned by
 *    the caller (pg_malloc/pg_free are provided for hosts without a HIP allocator).
 *  - the matrix A is a library-owned object (pg_mat): dense column-major like a Julia
 *    `Matrix{T}`, stored with the leading dimension padded to 1 KiB so that every column is
 *    16-byte aligned for the streaming kernels.
 *  - all work is enqueued on the context's HIP stream; an entry point that returns a scalar
 *    through a HOST pointer synchronises that stream before returning, the others are
 *    asynchronous.  One host thread per context at a time.
 *  - scalars cross the ABI as double; kernels compute in `dtype` with fp64 final reductions.
 *  - multi-GPU: rows of A are sharded one shard per process/GPU.  Register a SUM all-reduce
 *    with pg_ctx_set_allreduce(); pg_ls_* then reduce [grad ; f] (n+1 elements) once per
 *    gradient evaluation and 1 element per f-only evaluation.
 */
#ifndef PROXGRAD_HIP_H
#define PROXGRAD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an exported signature, a struct layout the host sees, or the set of exports changes (round 5: 2, then 3 with pg_mat_fused_tn_trio; round 6: 4 with pg_ctx_row_team_tune / _geometry / pg_mat_mul_multi --
 * rounds 3 and 4 added exports and fields under version 1).  Hosts compare pg_abi_version() with the value THEY were written against at
 * load time (Python: _lib.load; Julia: __init__) and refuse a stale or mismatched build with one clear message instead of a
 * missing symbol at some later call -- PG_LIB_PATH / PROXGRAD_HIP_LIB make pointing at another build easy. */
#define PG_ABI_VERSION 4

typedef int32_t pg_status;
enum {
  PG_OK = 0,
  PG_ERR_INVALID = -1,   /* bad argument (null pointer, negative size, unknown enum)        */
  PG_ERR_HIP = -2,       /* a HIP runtime call failed; see pg_last_error()                   */
  PG_ERR_ALLOC = -3,     /* device allocation failed                                         */
  PG_ERR_UNSUPPORTED = -4,
  PG_ERR_COLLECTIVE = -5, /* the registered all-reduce callback reported a failure           */
  PG_ERR_TIMEOUT = -6     /* a bounded wait inside a kernel expired: the workgroup teams of the long-column sweep wait
                           * for each other and one of them never arrived (see pg_ls_fused_pass).  The outputs of the call
                           * are undefined, its inputs intact; the iterators (pg_iter_*) recover by themselves           */
};

enum { PG_F32 = 0, PG_F64 = 1 };

/* proximable term g -- ProximalOperators.{NormL1, IndBox}, ProximalCore.Zero               */
enum { PG_G_ZERO = 0, PG_G_NORML1 = 1, PG_G_INDBOX = 2, PG_G_SQRNORML2 = 3 /* lam/2 ||.||^2: pg_mat_fused_dys only */ };

/* extrapolation sequences -- src/accel/nesterov.jl                                          */
enum {
  PG_SEQ_ADAPTIVE = 0, /* AdaptiveNesterovSequence(mf)   nesterov.jl:56-103 (FFB default)   */
  PG_SEQ_FIXED = 1,    /* FixedNesterovSequence          nesterov.jl:14-17                   */
  PG_SEQ_SIMPLE = 2,   /* SimpleNesterovSequence         nesterov.jl:36                      */
  PG_SEQ_CONSTANT = 3, /* ConstantNesterovSequence(m, s) nesterov.jl:51-54                   */
  PG_SEQ_HOST = 4,     /* coefficient supplied by the host each step (custom iterators)      */
  PG_SEQ_REPEATED = 5  /* the constant seq_p0 every step: Iterators.repeated(beta), which is what
                        * ConstantNesterovSequence returns (nesterov.jl:51-54)               */
};

/* flags reported in pg_iter_scalars.flags */
enum {
  PG_FLAG_GAMMA_TOO_SMALL = 1, /* fb_tools.jl:59-61 (@warn, not an error) */
  PG_FLAG_SWEEP_FALLBACK = 2,  /* this iteration's single sweep was refused at launch or timed out (PG_ERR_TIMEOUT): its
                                * uncommitted outputs were discarded and the iteration redone with two sweeps (A x, A' r) --
                                * same iterate, more reads of A (pg_iter_scalars.a_passes shows them) */
  PG_FLAG_COOP_SLOW = 4        /* reported once: the cooperative launch of the long-column sweep ran at under 4 TB/s twice in a
                                * row: the device is shared with a process that holds a cooperative queue and alternates between
                                * the two (same results, ~0.45 of the rate; a line on stderr names the remedy, PG_TN_TEAM_PLAIN=1) */
};

typedef struct pg_ctx pg_ctx;
typedef struct pg_mat pg_mat;
typedef struct pg_ls pg_ls;
typedef struct pg_iter pg_iter;
typedef struct pg_lbfgs pg_lbfgs;

/* SUM all-reduce over the row shards, in place on a device buffer, ordered on `stream`.
 * Returns 0 on success. */
typedef int (*pg_allreduce_fn)(void* user, void* dev_buf, int64_t count, int32_t dtype, void* stream);
/* Make `stream` wait for every all-reduce issued through the asynchronous begin callback.  Returns 0 on success. */
typedef int (*pg_allreduce_wait_fn)(void* user, void* stream);

typedef struct pg_device_info {
  int32_t device;
  int32_t compute_units;
  int32_t wavefront_size;
  int32_t lds_bytes_per_cu;
  int64_t global_mem_bytes;
  int32_t clock_khz;
  char arch[64];
  char name[128];
} pg_device_info;

/* ------------------------------------------------------------------ context / memory ---- */
int32_t pg_abi_version(void);
const char* pg_last_error(void);
/* `stream` is a hipStream_t (NULL = the device's default stream); it is borrowed, not owned. */
pg_status pg_ctx_create(int32_t device, void* stream, pg_ctx** out);
pg_status pg_ctx_destroy(pg_ctx* ctx);
pg_status pg_ctx_set_stream(pg_ctx* ctx, void* stream);
pg_status pg_ctx_set_allreduce(pg_ctx* ctx, pg_allreduce_fn fn, void* user);
/* Optional asynchronous pair: `begin` issues a SUM all-reduce that is ordered after the work already enqueued on
 * `stream` but does NOT block later work on it (e.g. ncclAllReduce on a side stream behind an event); `wait` makes
 * `stream` wait for all of them.  With the pair registered, a gradient evaluation runs A'r in column chunks and
 * overlaps each chunk's collective with the next chunk's pass (only the last chunk's collective is exposed). */
pg_status pg_ctx_set_allreduce_async(pg_ctx* ctx, pg_allreduce_fn begin, pg_allreduce_wait_fn wait, void* user);
/* Native collective (no host callback): RCCL is bound at run time with dlopen.  Rank 0 calls
 * pg_comm_get_unique_id and ships the PG_COMM_ID_BYTES bytes to every rank by any means (a file, MPI, sockets,
 * torch.distributed ...); every rank then calls pg_ctx_comm_init, which creates the communicator (ncclCommInitRank)
 * and installs the library's own all-reduce -- blocking form on the context stream, and with overlap != 0 also the
 * asynchronous pair on a side stream behind events (chunked pass T, see pg_ctx_set_allreduce_async). */
#define PG_COMM_ID_BYTES 128
int32_t pg_comm_available(void); /* 1 when librccl could be loaded, else 0 (then the two calls below return PG_ERR_UNSUPPORTED) */
pg_status pg_comm_get_unique_id(void* id_out /* PG_COMM_ID_BYTES */);
/* One communicator per context for the life of the job: a further call with the same (nranks, rank) keeps it, re-installs the
 * library's all-reduce and only switches the overlap mode (`id` is then ignored). */
pg_status pg_ctx_comm_init(pg_ctx* ctx, const void* id, int32_t nranks, int32_t rank, int32_t overlap);
pg_status pg_ctx_comm_destroy(pg_ctx* ctx);
/* telemetry of the native path: all-reduces issued so far and the sum of their lengths (elements) */
pg_status pg_ctx_comm_stats(pg_ctx* ctx, int64_t* calls, int64_t* elements);
/* Column sharding (the alternative to row sharding named in SURVEY 8(e)): rank `rank` of `nranks` holds the column block
 * A[:, J_rank] and the J_rank slices of all n-vectors; b and the residual are replicated.  The registered collective
 * (pg_ctx_set_allreduce / pg_ctx_comm_init) then carries A x (m elements) plus 8 * (nranks + 1) scalar slots (four scalars per rank as hi / lo pairs of
 * the working precision, and one shared group whose first word sums the ranks' sweep-timeout flags so that all ranks fall back together) -- ONE all-reduce
 * per iteration -- and A' r needs none, so the single-sweep iteration (pg_iter_opts.single_sweep) keeps working on
 * every rank (fixed step, or FastForwardBackward's adaptive step with reuse_residual).  nranks = 0: row sharding. */
pg_status pg_ctx_set_column_sharding(pg_ctx* ctx, int32_t nranks, int32_t rank);
/* Row teams: north_star's ROW layout (GPU p holds the row block A_p; benchmark/benchmarks.jl:15-16 become A_p x - b_p and
 * sum_p A_p' res_p) at ONE read of A per iteration.  Instead of all-reducing A' res between two sweeps, the devices exchange
 * the per-column partial dots INSIDE the sweep: every device pushes its partial, as a tagged 8-byte granule, into the inbox
 * of every device (peer-visible memory, xGMI stores), finds all partials of a column in its own memory a few steps later,
 * sums them in device order and goes on with the prox and A_p v while the column tile waits in LDS (csrc/pg_gemv_tn4.hip).
 * f and the bounded-wait timeout flag are exchanged the same way after the sweep, so a steady-state iteration issues no
 * collective; initialisation, the line search and the two-sweep fallback still use the registered all-reduce.
 *   _alloc   this context's inbox (fine-grained device memory, zeroed; freed with the context -- the peers write into it
 *            during their sweeps, so a context of a team is destroyed only after every device of the team has synchronised
 *            (pg_ctx_sync) and left the team or stopped iterating: a barrier before pg_ctx_destroy)
 *   _export  its IPC handle (64 bytes) for the other processes of the node;  _import  opens a peer's handle here
 *   pg_ctx_set_row_team(ctx, nranks, rank, inboxes, max_workgroups): inboxes[q] = device q's inbox as mapped into THIS
 *            process (inboxes[rank] = the own one); max_workgroups = workgroups per device (0: as many per compute unit as the
 *            geometry's parked tiles allow; -k: that number divided by k, for k members sharing ONE device -- tests), the same
 *            on every device; nranks <= 1 switches the mode off.  2..16 devices, row blocks (of equal or unequal length: the
 *            devices agree on the longest once per matrix and size the sweep for it) of at most 16384
 *            (Float32) / 8192 (Float64) rows per device; fixed step, or FastForwardBackward's adaptive step with
 *            reuse_residual (the line search's rejected trials use the registered all-reduce). */
pg_status pg_ctx_row_team_alloc(pg_ctx* ctx, void** inbox_out, int64_t* bytes_out);
pg_status pg_ctx_row_team_export(pg_ctx* ctx, void* handle_out /* 64 bytes */);
pg_status pg_ctx_row_team_import(pg_ctx* ctx, const void* handle /* 64 bytes */, void** inbox_out);
pg_status pg_ctx_set_row_team(pg_ctx* ctx, int32_t nranks, int32_t rank, void* const* inboxes, int32_t max_workgroups);
/* telemetry since the last pg_ctx_set_row_team (syncs): row-team sweeps launched; waves that did not find a step's granules at
 * their first look; polls (one per ~64 clocks) those waves spent waiting.  late_waves / (sweeps * steps * waves) near 0 means the
 * exchange fits its lag; a large wait_polls with no fallback means the fabric's latency, not the kernel, sets the rate. */
pg_status pg_ctx_row_team_stats(pg_ctx* ctx, int64_t* sweeps, int64_t* late_waves, int64_t* wait_polls);
/* One scalar exchange through the inboxes, to be called by every device of the team at the same point (after
 * pg_ctx_set_row_team and a barrier): device p contributes p + 1, *sum_out must come back as N (N + 1) / 2 on every device.
 * PG_ERR_TIMEOUT when a peer's granules never became visible here (bounded wait): the sweeps would fall back every time. */
pg_status pg_ctx_row_team_selftest(pg_ctx* ctx, double* sum_out);
/* The row-team sweep's geometry, per context and at run time (no rebuild, no PG_TUNE): the knobs a first run on real fabric turns.
 * Every device of the team sets the same values, before the sweep they should apply to; 0 = the library's choice.
 *   "PAIR"  1: ONE post per TWO steps -- a step's granules wait for the next step's and leave together, 16 * C bytes per inbox
 *              instead of two writes of 8 * C: half the fabric transactions, one step less of hand-off slack (2: back to one
 *              post per step); blocks up to 2048 (Float32) / 1024 (Float64) rows, where one wave holds the column
 *   "C"     columns per step (1 / 2 / 4: more columns = fewer, larger writes; only what is instantiated for the block length)
 *   "LAG"   tiles that wait in LDS for their totals;  "LAGR"  value - 1 tiles that wait in registers (1: none)
 *           -- the slack a granule has to arrive is (LAG + LAGR) steps of 16 KiB per wave
 *   "PF"    tiles in flight (1 / 2);  "WGS" workgroups per compute unit;  "W" waves per column (1 / 2 / 4)
 *   "K1"    1: the one-wave sweep of round 6 (default where it applies), 2: round 5's kernel
 *   "AHEAD" 1 (default): a step's granules are looked at one step before their use (the poll then returns with a tile that is waited
 *           for anyway: two tiles stay in flight), 2: at the start of the step that uses them (one step more of hand-off slack, half
 *           the streaming depth: what a fabric slower than LAG + LAGR - 1 steps might prefer)
 *   "SPIN"  the bounded wait, in polls of ~64 clocks (default 2^21, about 0.2 s): after it a wave gives up, the step is redone
 *           with two sweeps and the all-reduce, on every device
 * A combination without an instantiation is refused at the next sweep (PG_ERR_UNSUPPORTED: the iterator stays on two sweeps).
 * pg_ctx_row_team_geometry writes what the LAST row-team sweep of this context ran with, e.g.
 * "W=1 U=8 C=2 LAG=2 LAGR=2 PF=2 WGS=4 K1=1 PAIR=0 AHEAD=1 SPIN=2097152 WG=1024" ("none" before the first one). */
pg_status pg_ctx_row_team_tune(pg_ctx* ctx, const char* key, int64_t value);
pg_status pg_ctx_row_team_geometry(pg_ctx* ctx, char* buf, int64_t buflen);
pg_status pg_ctx_sync(pg_ctx* ctx);
pg_status pg_ctx_device_info(pg_ctx* ctx, pg_device_info* out);
/* Kernel timing with HIP events on the context's stream (bench.py's roofline leg).  While enabled, every
 * launch of the kernels below is bracketed by an event pair; pg_ctx_profile_read synchronises the stream and
 * returns the launch count and summed duration since the last reset. */
enum {
  PG_K_GEMV_N = 0, PG_K_GEMV_N_FINISH = 1, PG_K_GEMV_T = 2, PG_K_EPILOGUE = 3, PG_K_EXTRAPOLATE = 4, PG_K_DR_STEP = 5,
  PG_K_GEMV_TN = 6, PG_K_COUNT = 7
};
pg_status pg_ctx_profile_enable(pg_ctx* ctx, int32_t enable);
/* Restrict the event pairs to a set of kernels (bit k = kernel k of the enum above; default all): every pair is a
 * marker packet on the stream, so timing only the kernels of interest perturbs a short iteration less. */
pg_status pg_ctx_profile_select(pg_ctx* ctx, uint32_t kernel_mask);
pg_status pg_ctx_profile_reset(pg_ctx* ctx);
pg_status pg_ctx_profile_read(pg_ctx* ctx, int32_t kernel, int64_t* launches, double* total_ms);

/* Julia `Array` alloc/copy idioms (copy, similar, zero, copyto!) */
pg_status pg_malloc(pg_ctx* ctx, size_t bytes, void** dptr);
pg_status pg_free(pg_ctx* ctx, void* dptr);
pg_status pg_memcpy_h2d(pg_ctx* ctx, void* dst, const void* src, size_t bytes);
pg_status pg_memcpy_d2h(pg_ctx* ctx, void* dst, const void* src, size_t bytes); /* syncs */
pg_status pg_memcpy_d2d(pg_ctx* ctx, void* dst, const void* src, size_t bytes);
pg_status pg_memset_zero(pg_ctx* ctx, void* dst, size_t bytes);

/* ------------------------------------------------------------------ matrix A ------------ */
/* Julia `Matrix{T}` (m x n, column-major) held by `LeastSquares(A, b)`:
 * benchmark/benchmarks.jl:41-52 ; test/problems/test_lasso_small.jl:17-23,37 */
pg_status pg_mat_create(pg_ctx* ctx, int32_t dtype, int64_t m, int64_t n, pg_mat** out);
pg_status pg_mat_destroy(pg_mat* A);
pg_status pg_mat_upload(pg_mat* A, const void* host_colmajor, int64_t ld_host);
pg_status pg_mat_set_from_device(pg_mat* A, const void* dev_colmajor, int64_t ld_dev);
pg_status pg_mat_download(pg_mat* A, void* host_colmajor, int64_t ld_host); /* syncs */
/* synthetic instance of SURVEY 8(d): A[i,j] = ih8(seed, row_offset+i, j) * scale, identical
 * bit-for-bit to oracle/proxgrad_oracle.py::synthetic_matrix for every row shard */
pg_status pg_mat_generate(pg_mat* A, uint32_t seed, int64_t row_offset, double scale);
/* ... the block at (row_offset, col_offset) of the same global matrix (row shards use the first, column shards the second) */
pg_status pg_mat_generate_block(pg_mat* A, uint32_t seed, int64_t row_offset, int64_t col_offset, double scale);
pg_status pg_mat_info(const pg_mat* A, int64_t* m, int64_t* n, int64_t* ld, int32_t* dtype, void** dptr);
/* y = A x  (mul!(y, A, x)) and g = A' r  (mul!(g, A', r)) -- the two GEMV orientations on the
 * column-major store; used by LeastSquares and (later) PANOC's `mul!` with A: panoc.jl:150-190 */
pg_status pg_mat_mul(pg_mat* A, const void* x, void* y);
/* ys[k] = A xs[k] for nv <= 3 vectors on ONE read of A, each ys[k] bit-identical to pg_mat_mul's (the same multiply-adds in the same
 * order): the step-size search of src/utilities/fb_tools.jl:46-55 forms `mul!(Az, A, z)` once per halving of gamma, and its next
 * candidates gamma / 2, gamma / 4, gamma / 8 differ in z only -- their images are taken together, the decisions stay the reference's.
 * PG_ERR_UNSUPPORTED for sharded operators and below 13 row groups (3073 Float32 / 1537 Float64 rows): one product at a time there. */
pg_status pg_mat_mul_multi(pg_mat* A, int32_t nv, const void* const* xs, void* const* ys);
pg_status pg_mat_mul_adjoint(pg_mat* A, const void* r, void* g);
/* The single sweep for x -> f(A x) compositions (PANOC: panoc.jl:186, :199-201, and the `mul!(Az, A, z)` of the next line
 * search, fb_tools.jl:43): for a caller-supplied m-vector r (= grad f(A x)),
 *   At_r = A' r ; y = x - gamma At_r ; z = prox_{gamma g}(y) ; res = x - z ; Az = A z
 * in ONE read of A.  scalars_out (host, may be NULL) = { g(z), norm(res, Inf), dot(At_r, res), norm(res)^2 }.
 * Unsharded matrices with m <= 262144 (f32) / 131072 (f64) rows (columns longer than 32768 / 16384 rows are split over
 * teams of workgroups); PG_ERR_UNSUPPORTED otherwise. */
pg_status pg_mat_fused_tn(pg_mat* A, const void* r, const void* x, double gamma, int32_t g_kind, double g_p0, double g_p1,
                          void* At_r, void* y, void* z, void* res, void* Az, double* scalars_out);
/* The same sweep leaving Ares = A (x - z), the image of the forward-backward residual, instead of A z: the residual's image
 * as a PRODUCT of the (small) residual -- its error scales with norm(res), where A x - A z carries eps * norm(A x) however
 * small the residual has become.  PANOC with the L-BFGS image slab (pg_lbfgs_images_*) runs on it: d = -H res (panoc.jl:
 * 114-117), A d from the images of A res, A y = A res+ - A res (panoc.jl:122-126), and the line search's A z = A x - A res
 * (fb_tools.jl:43) shares its rounding with the A x that f(A x) was taken at. */
pg_status pg_mat_fused_tn_res(pg_mat* A, const void* r, const void* x, double gamma, int32_t g_kind, double g_p0, double g_p1,
                              void* At_r, void* y, void* z, void* res, void* Ares, double* scalars_out);
/* TWO instances of pg_mat_fused_tn in ONE read of A: the same gamma and g, two pairs (r, x), every output twice.  ZeroFPR's line
 * search (zerofpr.jl:200-217: x = xbar_prev + tau d; A' grad f(A x); y; xbar = prox(y); res; and A xbar for the next iteration,
 * :167) evaluates its trial points one sweep each; with the points of tau and tau / 2 carried through the same pass a rejected
 * first trial costs no second read of A.  Each instance's results equal pg_mat_fused_tn's to the last bits of the working
 * precision (eight waves share a column here, four there: the same fma chains, another grouping of the partial sums).
 * scalars_out (host, may be NULL): the four scalars of the first instance, then of the second.
 * Columns of 33 .. 64 row groups of 1 KiB (8193 .. 16384 rows in Float32: BASELINE config 4's 16384; 4097 .. 8192 in Float64);
 * PG_ERR_UNSUPPORTED otherwise (the caller falls back to one trial point per sweep). */
pg_status pg_mat_fused_tn_pair(pg_mat* A, const void* r1, const void* x1, const void* r2, const void* x2, double gamma, int32_t g_kind,
                               double g_p0, double g_p1, void* At_r1, void* y1, void* z1, void* res1, void* Az1, void* At_r2, void* y2,
                               void* z2, void* res2, void* Az2, double* scalars_out);
/* The same with the images of the residuals, Ares_k = A (x_k - z_k), in place of A z_k (cf. pg_mat_fused_tn_res).  PANOCplus uses it to
 * fold its second pass over A -- `mul!(state.At_grad_f_Az, adjoint(iter.A), state.grad_f_Az)` (panocplus.jl:225), needed by the
 * stopping criterion alone (:243) -- into the FIRST sweep of the next iteration (:199-210 at tau = 1), taken speculatively: one
 * read of A per iteration instead of two. */
pg_status pg_mat_fused_tn_pair_res(pg_mat* A, const void* r1, const void* x1, const void* r2, const void* x2, double gamma, int32_t g_kind,
                                   double g_p0, double g_p1, void* At_r1, void* y1, void* z1, void* res1, void* Ares1, void* At_r2, void* y2,
                                   void* z2, void* res2, void* Ares2, double* scalars_out);
/* THREE instances of pg_mat_fused_tn in ONE read of A (gemv_tnm_trio_kernel): the trial points tau, tau / 2 and tau / 4 of
 * zerofpr.jl:200-217 through the same pass.  r, x: three device pointers each (host arrays of pointers); At_r, y, z, res, Az: the
 * outputs of instance k at index k; image_of_res != 0: Az[k] = A (x_k - z_k).  scalars_out (host, may be NULL): twelve values, the
 * four of pg_mat_fused_tn per instance.  Same column lengths as pg_mat_fused_tn_pair; PG_ERR_UNSUPPORTED otherwise. */
pg_status pg_mat_fused_tn_trio(pg_mat* A, const void* const r[3], const void* const x[3], double gamma, int32_t g_kind, double g_p0,
                               double g_p1, void* const At_r[3], void* const y[3], void* const z[3], void* const res[3],
                               void* const Az[3], int32_t image_of_res, double* scalars_out);
/* ------------------------------------------------------------------ LeastSquares -------- */
/* f(x) = lam/2 ||A x - b||^2 -- ProximalOperators.LeastSquares(A, b[, lam]) with the
 * value_and_gradient method of benchmark/benchmarks.jl:11-17.  `b` is a device m-vector
 * borrowed for the lifetime of the object. */
pg_status pg_ls_create(pg_ctx* ctx, pg_mat* A, const void* b, double lam, pg_ls** out);
pg_status pg_ls_destroy(pg_ls* f);
/* (f(x), grad f(x)) = (||A x - b||^2 / 2, A'(A x - b))   benchmarks.jl:15-16 ; called from
 * forward_backward.jl:67,113 ; fast_forward_backward.jl:75,138 ; fb_tools.jl:10,44,53 */
pg_status pg_ls_value_and_gradient(pg_ls* f, const void* x, void* grad_out, double* f_out);
/* f(x) only (the line search discards the gradient in FFB: fast_forward_backward.jl:110-129) */
pg_status pg_ls_value(pg_ls* f, const void* x, double* f_out);
/* ProximalCore.gradient!(y, f, x) -> f(x)  (ProximalCore <= 0.1 callers; same arithmetic) */
pg_status pg_ls_gradient(pg_ls* f, void* grad_out, const void* x, double* f_out);
/* r = A x - b of the last evaluation (device m-vector, library-owned; valid until next call) */
pg_status pg_ls_residual_ptr(pg_ls* f, const void** r_out);
/* ONE sweep over A for a whole proximal-gradient iteration (benchmarks.jl:16 `f.A' * res`, forward_backward.jl:117-120 /
 * fast_forward_backward.jl:140-142, then :135 and benchmarks.jl:15 of the NEXT iteration): with the residual held by f
 * being that of x (last pg_ls_value / pg_ls_value_and_gradient / pg_ls_fused_pass evaluation),
 *   grad = lam A' r ; y = x - gamma grad ; z_new = prox_{gamma g}(y) ; res = x - z_new ;
 *   v_next = z_new + beta (z_new - z_old) ; f's residual := A v_next - b.
 * A column's contribution to A v_next is accumulated while the column is still in registers, so A is read once.
 * Unsharded or column-sharded operators with m <= 262144 (f32) / 131072 (f64) rows; PG_ERR_UNSUPPORTED otherwise.
 * scalars_out (host, may be NULL) = { f(v_next), g(z_new), norm(res, Inf), dot(grad, res), norm(res)^2 }.
 * Columns longer than one workgroup's registers (> 32768 rows f32) are split over TEAMS of workgroups that exchange their
 * partial dot products while the kernel runs; the launch is cooperative, so either every team member is resident or the
 * launch is refused (PG_ERR_UNSUPPORTED, nothing written).  One such sweep at a time per device: a member that still has
 * not heard from its team after a bounded wait gives up, and the call (or, with scalars_out == NULL, the next call that
 * synchronises) returns PG_ERR_TIMEOUT -- grad, y, z_new, res, v_next and f's residual are then undefined, x and z_old
 * intact; re-evaluate f at x (pg_ls_value) before trying again, or use pg_ls_value_and_gradient + pg_fb_epilogue.
 * On a device shared with ANOTHER PROCESS that has used a cooperative launch (even an idle one) the cooperative team sweep
 * runs at about half its rate: the device does not run the cooperative queues of two processes side by side.  The
 * environment variable PG_TN_TEAM_PLAIN=1 (read when the context is created) launches the teams plainly instead -- full
 * rate, no residency guarantee, the bounded wait and PG_ERR_TIMEOUT as the safety net. */
pg_status pg_ls_fused_pass(pg_ls* f, const void* x, const void* z_old, double gamma, double beta, int32_t g_kind,
                           double g_p0, double g_p1, void* grad, void* y, void* z_new, void* res, void* v_next,
                           double* scalars_out);

/* ------------------------------------------------------------------ prox operators ------ */
/* ProximalCore.prox!(y, g::NormL1, x, gamma) -> g(y) : y_i = sign(x_i) max(|x_i| - gamma lam, 0),
 * returns lam ||y||_1.  Call sites forward_backward.jl:72,118 ; fast_forward_backward.jl:80,141 ;
 * fb_tools.jl:49.  y may alias x.  gy_out may be NULL (no sync, value skipped). */
pg_status pg_prox_norml1(pg_ctx* ctx, int32_t dtype, int64_t n, void* y, const void* x, double lam,
                         double gamma, double* gy_out);
/* prox!(y, g::IndBox, x, gamma) -> 0 : y_i = min(hi_i, max(lo_i, x_i)); scalar bounds lo/hi are used
 * where lo_vec / hi_vec are NULL.  test/problems/test_nonconvex_qp.jl:19,33 */
pg_status pg_prox_indbox(pg_ctx* ctx, int32_t dtype, int64_t n, void* y, const void* x, double lo,
                         double hi, const void* lo_vec, const void* hi_vec, double* gy_out);
/* g(x) for NormL1 (lam ||x||_1) */
pg_status pg_norml1_value(pg_ctx* ctx, int32_t dtype, int64_t n, const void* x, double lam, double* out);
/* NormL1 with per-element weights (ProximalOperators.NormL1(lambda::AbstractArray)), lam_vec a device n-vector:
 * y_i = sign(x_i) max(|x_i| - gamma lam_i, 0), returns sum_i lam_i |y_i| ; and the value sum_i lam_i |x_i|. */
pg_status pg_prox_norml1w(pg_ctx* ctx, int32_t dtype, int64_t n, void* y, const void* x, const void* lam_vec,
                          double gamma, double* gy_out);
pg_status pg_norml1w_value(pg_ctx* ctx, int32_t dtype, int64_t n, const void* x, const void* lam_vec, double* out);

/* ------------------------------------------------------------------ BLAS-1 / broadcasts -- */
/* out .= a .* x .+ b .* y   (y may be NULL when b == 0); covers `y .= x .- gamma .* grad`
 * (forward_backward.jl:117), `res .= x .- z` (:120), `x .+ 1` via pg_add_scalar */
pg_status pg_axpby(pg_ctx* ctx, int32_t dtype, int64_t n, void* out, double a, const void* x, double b,
                   const void* y);
pg_status pg_add_scalar(pg_ctx* ctx, int32_t dtype, int64_t n, void* out, const void* x, double c);
pg_status pg_fill(pg_ctx* ctx, int32_t dtype, int64_t n, void* out, double c);
/* x .= z .+ beta .* (z .- z_prev)   fast_forward_backward.jl:135 */
pg_status pg_extrapolate(pg_ctx* ctx, int32_t dtype, int64_t n, void* x, const void* z, const void* z_prev,
                         double beta);
pg_status pg_dot(pg_ctx* ctx, int32_t dtype, int64_t n, const void* x, const void* y, double* out);
pg_status pg_nrm2sq(pg_ctx* ctx, int32_t dtype, int64_t n, const void* x, double* out); /* norm(x)^2 */
pg_status pg_nrminf(pg_ctx* ctx, int32_t dtype, int64_t n, const void* x, double* out); /* norm(x, Inf) */
/* fused forward-backward epilogue: y = x - gamma grad ; z = prox_{gamma g}(y) ; res = x - z and the
 * four reductions the iteration needs: scalars_out = { g(z), ||res||_inf, <grad,res>, ||res||^2 }
 * (forward_backward.jl:117-120 + fb_tools.jl:3-5 + forward_backward.jl:125-126).
 * g_kind in {PG_G_ZERO, PG_G_NORML1, PG_G_INDBOX}; g_p0 = lam | lo ; g_p1 = hi. */
pg_status pg_fb_epilogue(pg_ctx* ctx, int32_t dtype, int64_t n, const void* x, const void* grad, double gamma,
                         int32_t g_kind, double g_p0, double g_p1, void* y, void* z, void* res,
                         double* scalars_out /* host, 4 doubles; NULL = leave on device */);

/* ------------------------------------------------------------------ smooth losses (config 4) ---- */
/* The smooth term f of PANOC's  minimize f(A x) + g(x)  (src/algorithms/panoc.jl:39-52), evaluated on an
 * m-vector u = A x:  value_and_gradient(f, u) -> (f(u), grad f(u))   (panoc.jl:90,182,216,240 ; fb_tools.jl:44).
 *   PG_LOSS_SQDIST   f(u) = ||u - b||^2 / 2              benchmark/benchmarks.jl:19-28 (SquaredDistance)
 *   PG_LOSS_LOGISTIC f(u) = sum(log.(1 .+ exp.(-(u .- b))))  test/problems/test_sparse_logistic_small.jl:20-26 */
enum { PG_LOSS_SQDIST = 0, PG_LOSS_LOGISTIC = 1 };
pg_status pg_loss_value_and_gradient(pg_ctx* ctx, int32_t dtype, int32_t loss, int64_t m, const void* u,
                                     const void* b, void* grad, double* f_out);

/* ------------------------------------------------------------------ Douglas-Rachford (config 3) --- */
/* Separable quadratic f(x) = sum_i d_i x_i^2 / 2 + q_i x_i  (ProximalOperators Tilt(SqrNormL2(d), q), i.e.
 * Quadratic(Diagonal(d), q)): prox!(y, f, x, gamma) -> f(y), y_i = (x_i - gamma q_i) / (1 + gamma d_i).
 * d / q are used where d_vec / q_vec are NULL. */
pg_status pg_prox_sepquad(pg_ctx* ctx, int32_t dtype, int64_t n, void* y, const void* x, const void* d_vec, double d,
                          const void* q_vec, double q, double gamma, double* fy_out);
/* One DouglasRachfordIteration step, fused into a single HBM sweep (src/algorithms/douglas_rachford.jl:53-63):
 *   prox!(y, f, x, gamma); r .= 2 .* y .- x; prox!(z, g, r, gamma); res .= y .- z; x .-= res
 * x is updated in place, y (the solution, :70) is always written; r, z, res may be NULL (not materialised).
 * scalars_out = { norm(res, Inf), f(y), g(z) } (stop rule :65-69: norm(res, Inf) / gamma <= tol). */
pg_status pg_dr_step(pg_ctx* ctx, int32_t dtype, int64_t n, void* x, void* y, void* r, void* z, void* res,
                     const void* d_vec, double d, const void* q_vec, double q, int32_t g_kind, double g_p0,
                     double g_p1, double gamma, double* scalars_out /* host, 3 doubles; NULL = no sync */);

/* Stepping (douglas_rachford.jl:53-63, one Base.iterate per call) with the NEXT iteration already in flight: iteration k + 1
 * reads nothing but x_k, so it can be launched -- out of place, into a second set of state vectors -- before the host has
 * read iteration k's norm(res, Inf); the host's round trip (half as long as the 34 us kernel at n = 10^7) then hides behind
 * the kernel.  _async: one iteration x_in -> (x_out, y, r, z, res) (r, z, res nullable), scalars to slot 0 | 1, an event
 * recorded behind it; _wait: block on THAT iteration only and return { norm(res, Inf), f(y), g(z) }. */
pg_status pg_dr_step_async(pg_ctx* ctx, int32_t dtype, int64_t n, const void* x_in, void* x_out, void* y, void* r, void* z,
                           void* res, const void* d_vec, double d, const void* q_vec, double q, int32_t g_kind, double g_p0,
                           double g_p1, double gamma, int32_t slot);
pg_status pg_dr_step_wait(pg_ctx* ctx, int32_t slot, double* scalars_out /* host, 3 doubles */);
/* The DouglasRachford driver loop (src/ProximalAlgorithms.jl:114-123 with the default stop rule
 * norm(res, Inf) / gamma <= tol, douglas_rachford.jl:65-69, evaluated in T) inside the library.  With block = 8, 16, 32 or 64
 * that many iterations run per HBM sweep (f and g are separable, so the iterates of an element stay in registers;
 * the stop rule of every inner iteration is still evaluated and, when one of them fires, the block is replayed up to
 * it, so the state left behind is bit-identical to stepping with pg_dr_step).  A remainder of fewer than `block` iterations
 * before maxit runs in the next smaller block sizes, the last < 8 as single steps.  block = 1 steps one by one.
 * x_alt: caller-owned scratch n-vector (ping-pong partner of x; required when block > 1).  On return x, y (and r, z,
 * res when given) hold the state of iteration *k_out; scalars_out as in pg_dr_step. */
pg_status pg_dr_run(pg_ctx* ctx, int32_t dtype, int64_t n, void* x, void* x_alt, void* y, void* r, void* z, void* res,
                    const void* d_vec, double d, const void* q_vec, double q, int32_t g_kind, double g_p0, double g_p1,
                    double gamma, double tol, int64_t maxit, int32_t block, int64_t* k_out,
                    double* scalars_out /* host, 3 doubles, may be NULL */);

/* ------------------------------------------------------------------ fused iterations ---- */
/* Options = the keyword arguments of ForwardBackwardIteration (forward_backward.jl:38-48) and
 * FastForwardBackwardIteration (fast_forward_backward.jl:44-56), f = LeastSquares, g by kind. */
typedef struct pg_iter_opts {
  int32_t fast;          /* 0 = ForwardBackward, 1 = FastForwardBackward                         */
  int32_t adaptive;      /* -1 = default (gamma <= 0 && Lf <= 0), else 0/1                        */
  double Lf;             /* <= 0: nothing                                                        */
  double gamma;          /* <= 0: nothing (then 1/Lf, or estimated: fb_tools.jl:7-12)             */
  double minimum_gamma;  /* 1e-7  */
  double reduce_gamma;   /* 0.5   */
  double increase_gamma; /* 1.0   */
  double mf;             /* FFB: convexity modulus (0)                                           */
  int32_t seq_kind;      /* FFB: PG_SEQ_*                                                        */
  double seq_p0, seq_p1; /* PG_SEQ_CONSTANT: (m, stepsize) ; PG_SEQ_REPEATED: seq_p0 = the coefficient    */
  int32_t g_kind;        /* PG_G_*                                                               */
  double g_p0, g_p1;     /* NormL1: lam | IndBox: lo, hi                                         */
  int32_t reuse_residual; /* FFB adaptive: 1 (default) = form A x - b at the extrapolated point from the residuals
                          * the line search already holds, (1+beta)(A z - b) - beta (A z_prev - b): 2 passes over A per
                          * iteration instead of 3 (the reference does 4); 0 = recompute A x like the reference */
  int32_t single_sweep;  /* 1 (default) = iterate with ONE read of A per iteration where the operator allows it
                          * (pg_ls_fused_pass: unsharded, m <= 262144 f32 / 131072 f64 rows; FB / FFB with a fixed step, FFB
                          * with the adaptive step and reuse_residual): the sweep that forms A' r also applies the prox
                          * to each finished column and accumulates the NEXT residual from it while it is in registers.
                          * Same iterates up to summation order.  0 = two sweeps (A x, then A' r) like the reference. */
} pg_iter_opts;

/* the scalar part of ForwardBackwardState / FastForwardBackwardState plus line-search telemetry */
typedef struct pg_iter_scalars {
  double gamma;    /* state.gamma                                                               */
  double f_x;      /* state.f_x                                                                 */
  double g_z;      /* state.g_z                                                                 */
  double res_inf;  /* norm(state.res, Inf)   (stop rule: res_inf / gamma <= tol)                */
  double beta;     /* last extrapolation coefficient (FFB)                                      */
  double f_z;      /* f(z) of the last accepted line-search trial (adaptive), else NaN          */
  double f_z_upp;  /* quadratic model value at the accepted trial (adaptive), else NaN          */
  int32_t n_backtracks; /* rejected trials in this step                                         */
  int32_t flags;        /* PG_FLAG_*                                                            */
  int64_t a_passes;     /* cumulative full reads of A since pg_iter_init (telemetry)            */
} pg_iter_scalars;

/* device vectors of the state (forward_backward.jl:52-63, fast_forward_backward.jl:60-71); the
 * pointers change across steps exactly where the reference swaps references (x <-> z, ...) */
typedef struct pg_iter_state {
  void* x;
  void* grad_f_x;
  void* y;
  void* z;
  void* res;
  void* z_prev;   /* FFB only, else NULL */
  void* grad_f_z; /* FB only, else NULL  */
} pg_iter_state;

pg_status pg_iter_opts_default(pg_iter_opts* opts);
/* ForwardBackwardIteration(; f, g, x0, ...) / FastForwardBackwardIteration(; ...) */
pg_status pg_iter_create(pg_ctx* ctx, pg_ls* f, const pg_iter_opts* opts, pg_iter** out);
pg_status pg_iter_destroy(pg_iter* it);
/* g = IndBox with PER-ELEMENT bounds (ProximalOperators.IndBox(lo::AbstractArray, hi::AbstractArray)): lo / hi are
 * device n-vectors (this rank's slices under column sharding), borrowed for the life of the iterator; call after
 * pg_iter_create (g_kind = PG_G_INDBOX; g_p0 / g_p1 are then ignored) and before pg_iter_init.  NULL, NULL: the scalars again.
 * g = NormL1 with PER-ELEMENT weights (ProximalOperators.NormL1(lambda::AbstractArray): g(x) = sum_j lam_j |x_j|, prox =
 * soft threshold by gamma lam_j): the weights go in `lo`, `hi` must be NULL (g_kind = PG_G_NORML1; g_p0 is then ignored).
 * The single sweep reads them as one or two more n-vector streams next to x and z_old.  pg_iter_run_small / _coop take
 * scalar parameters only (PG_ERR_UNSUPPORTED). */
pg_status pg_iter_set_g_vectors(pg_iter* it, const void* lo, const void* hi);
/* Base.iterate(iter): forward_backward.jl:65-84 / fast_forward_backward.jl:73-97.
 * x0 is a DEVICE n-vector; it is copied, never mutated (test_lasso_small.jl:54). */
pg_status pg_iter_init(pg_iter* it, const void* x0, pg_iter_scalars* out);
/* Base.iterate(iter, state): forward_backward.jl:86-123 / fast_forward_backward.jl:106-145.
 * host_beta is used only with PG_SEQ_HOST. */
pg_status pg_iter_step(pg_iter* it, double host_beta, pg_iter_scalars* out);
/* IterativeAlgorithm loop (src/ProximalAlgorithms.jl:114-123) with the default stopping rule:
 * runs until k >= maxit or res_inf/gamma <= tol; *k_out counts the init state like the reference.
 * Call after pg_iter_init; k_start is the current k (1 right after init). */
pg_status pg_iter_run(pg_iter* it, int64_t k_start, int64_t maxit, double tol, int64_t* k_out,
                      pg_iter_scalars* out);
/* Device-resident variant for fixed-step runs: `check_every` iterations are enqueued back to
 * back with no host synchronisation in between (nothing the host decides depends on them), then the scalar block is
 * read once and the stopping rule evaluated; k_out advances in steps of `check_every` (capped by maxit).  With
 * check_every = 1 this is pg_iter_run.  Not available with an adaptive step: backtracking is a host decision.
 * A long-column sweep that times out inside a batch (PG_ERR_TIMEOUT at the batch's read-back) cannot be redone -- the
 * iterations behind it are already enqueued -- so the call fails and the state is undefined: re-run pg_iter_init and use
 * pg_iter_run, which redoes a lost sweep per iteration (the Python algorithm object does exactly that). */
pg_status pg_iter_run_batched(pg_iter* it, int64_t k_start, int64_t maxit, double tol, int32_t check_every,
                              int64_t* k_out, pg_iter_scalars* out);
/* Launch-bound sizes (0 < m * n <= 2^20 elements, e.g. the reference's shipped benchmark instances): the whole
 * driver loop -- stop rule, line search, extrapolation sequence, both GEMV orientations, prox -- runs inside ONE
 * launch of one 1024-thread workgroup (A stays L2-resident); same control flow and state as pg_iter_run, no host
 * round trip per iteration.  The iterator can be stepped normally afterwards. */
pg_status pg_iter_run_small(pg_iter* it, int64_t k_start, int64_t maxit, double tol, int64_t* k_out,
                            pg_iter_scalars* out);
/* Cache-resident sizes (m <= 4096 f64 / 8192 f32 rows, A up to 256 MiB): the same driver loop in ONE cooperative
 * launch of `blocks` 1024-thread workgroups (<= one per CU; 0 = chosen from the size of A) that meet at grid
 * barriers: pass N as (row block, column slice) partial sums | barrier | every workgroup combines the residual into
 * its LDS, then A' r, prox and the scalar partials per column | barrier.  A fixed-step iteration costs two grid
 * barriers and no launch; the line search adds one per trial.  A stays L2-resident, the three residual vectors of
 * the adaptive FFB iteration live in LDS.  Barriers are bounded: a workgroup that never arrives makes the call
 * return an error instead of hanging.  Same control flow, state and follow-up use as pg_iter_run_small. */
pg_status pg_iter_run_coop(pg_iter* it, int64_t k_start, int64_t maxit, double tol, int32_t blocks, int64_t* k_out,
                           pg_iter_scalars* out);
pg_status pg_iter_state_view(pg_iter* it, pg_iter_state* out);
/* Checkpoint / resume.  In the reference `iterate(iter, saved_state)` continues a solve from any saved state, because the
 * state struct holds all algorithm memory (forward_backward.jl:52-63; fast_forward_backward.jl:60-71 with the mutable
 * AdaptiveNesterovSequence, nesterov.jl:56-60).  _download writes that memory -- state vectors, residual vectors, gamma,
 * f_x, g_z, the sequence state (stepsize, theta, t, k), and the speculative first half of the next single-sweep iteration
 * -- into ONE host blob of pg_iter_state_bytes bytes (syncs); _upload puts a blob into an iterator created with the same
 * options over an equal f (pg_iter_init not needed), after which pg_iter_step continues bit-identically to the solve the
 * blob was taken from.  Host-drawn extrapolation sequences (PG_SEQ_HOST) live in the caller and are refused. */
pg_status pg_iter_state_bytes(pg_iter* it, int64_t* bytes_out);
pg_status pg_iter_state_download(pg_iter* it, void* host_blob, int64_t bytes);
pg_status pg_iter_state_upload(pg_iter* it, const void* host_blob, int64_t bytes, pg_iter_scalars* out);

/* ------------------------------------------------------------------ L-BFGS (config 4) --- */
/* LBFGSOperator{M}: src/accel/lbfgs.jl:5-95 */
pg_status pg_lbfgs_create(pg_ctx* ctx, int32_t dtype, int32_t M, int64_t n, pg_lbfgs** out);
pg_status pg_lbfgs_destroy(pg_lbfgs* L);
pg_status pg_lbfgs_update(pg_lbfgs* L, const void* s, const void* y); /* update!  lbfgs.jl:30-50 */
pg_status pg_lbfgs_reset(pg_lbfgs* L);                               /* reset!   lbfgs.jl:52-55 */
pg_status pg_lbfgs_apply(pg_lbfgs* L, void* d, const void* v);       /* mul!     lbfgs.jl:64-95 */

/* Images of the stored pairs under a linear map A (m rows): with A s_i and A y_i kept next to s_i, y_i, the image of the
 * quasi-Newton direction, A (H v), follows from A v and the two-loop coefficients of the LAST pg_lbfgs_apply without
 * reading A:  A d = H0 (A v - sum alpha_i A y_i) + sum (alpha_i - beta_i) A s_i.  This removes the `mul!(state.Ad, iter.A,
 * state.d)` of src/algorithms/panoc.jl:180 (zerofpr.jl:193; the `mul!(state.Ax, iter.A, state.x)` of panocplus.jl:199):
 * v = res = x - z, A v = A x - A z are m-vectors the iteration already holds (panoc.jl:181 keeps A x, the forward-backward
 * sweep leaves A z), and the images of a new pair are differences of such m-vectors (A s = A x+ - A x, A y = A res+ - A res:
 * panoc.jl:122-126).  _enable(m) allocates the image slab (2 M m-vectors); _update(As, Ay) must follow every
 * pg_lbfgs_update with the images of the same pair (ignored when the pair was rejected, <s, y> <= 0: lbfgs.jl:34);
 * _apply(Ad, Av) must follow the pg_lbfgs_apply whose direction it maps and is refused (PG_ERR_INVALID) when a pair that
 * apply used has no current image; _ready says beforehand whether every pair the NEXT apply will use has one. */
pg_status pg_lbfgs_images_enable(pg_lbfgs* L, int64_t m);
pg_status pg_lbfgs_images_update(pg_lbfgs* L, const void* As, const void* Ay);
pg_status pg_lbfgs_images_apply(pg_lbfgs* L, void* Ad, const void* Av);
pg_status pg_lbfgs_images_ready(pg_lbfgs* L, int32_t* ready_out);
#ifdef __cplusplus
}
#endif
#endif /* PROXGRAD_HIP_H */
