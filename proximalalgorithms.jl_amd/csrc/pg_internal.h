// Internal definitions shared by the translation units of libproxgrad_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "proxgrad_hip.h"
#include "proxgrad_hip_ext.h"

// ---------------------------------------------------------------------------------------------
// error handling
// ---------------------------------------------------------------------------------------------
void pg_set_error(const char* fmt, ...);

#define PG_HIP(call)                                                                              \
  do {                                                                                            \
    hipError_t e__ = (call);                                                                      \
    if (e__ != hipSuccess) {                                                                      \
      pg_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__);   \
      return PG_ERR_HIP;                                                                          \
    }                                                                                             \
  } while (0)

#define PG_REQUIRE(cond, msg)                                  \
  do {                                                         \
    if (!(cond)) {                                             \
      pg_set_error("invalid argument: %s (%s)", msg, #cond);   \
      return PG_ERR_INVALID;                                   \
    }                                                          \
  } while (0)

#define PG_TRY(expr)                  \
  do {                                \
    pg_status s__ = (expr);           \
    if (s__ != PG_OK) return s__;     \
  } while (0)

#define PG_LAUNCH_CHECK()                                                                   \
  do {                                                                                      \
    hipError_t e__ = hipGetLastError();                                                     \
    if (e__ != hipSuccess) {                                                                \
      pg_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__,    \
                   __LINE__);                                                               \
      return PG_ERR_HIP;                                                                    \
    }                                                                                       \
  } while (0)

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
// scalar-block slots (device doubles)
enum {
  PG_S_F = 0,        // f = lam/2 ||Ax-b||^2 of the last residual evaluation
  PG_S_GZ = 1,       // g(z)
  PG_S_RESINF = 2,   // ||res||_inf
  PG_S_DOT = 3,      // <grad, res>
  PG_S_RESSQ = 4,    // ||res||^2
  PG_S_MISC = 5,     // dot / nrm2sq / nrminf / prox value results (2 slots)
  PG_S_DR = 8,       // Douglas-Rachford step: { ||res||_inf, f(y), g(z) }
  PG_S_FNEXT = 12,   // f at the speculative next point of the single-sweep iteration (2 slots, alternating)
  PG_S_TEAMERR = 14, // set to 1 by a workgroup team of the long-column sweep that gave up waiting for a member
  PG_S_DRRUN = 16,   // pg_dr_run block: { ||res||_inf of each of the K <= 64 inner iterations, f(y), g(z) }
  PG_S_DRRUN2 = 82,  // second set of the same (two blocks of pg_dr_run are in flight)
  PG_S_DRA = 148,    // pg_dr_step_async, slot 0: { ||res||_inf, f(y), g(z) }
  PG_S_DRB = 151,    // ... slot 1 (the iteration launched while slot 0's is being looked at)
  PG_S_PAIR = 154,   // pg_mat_fused_tn_pair: { g(z), ||res||_inf, <At_r, res>, ||res||^2 } of the first and of the second instance (pg_mat_fused_tn_trio: and of the third)
  PG_S_COUNT = 166
};

constexpr int PG_RED_MAX_BLOCKS = 4096;  // max grid of any kernel that uses grid_reduce_finalize
constexpr int PG_RED_MAX_NS = 66;

// RCCL communicator bound by pg_ctx_comm_init (csrc/pg_comm.hip)
struct pg_comm {
  void* comm = nullptr;  // ncclComm_t
  hipStream_t side = nullptr;
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  int nranks = 1, rank = 0;
  int64_t calls = 0, elements = 0;  // telemetry: all-reduces issued through the native path and their total length
};

// Row teams (pg_ctx_set_row_team): the devices of a row-sharded job exchange per-column partial dots through each other's
// inbox (peer-visible memory) inside the sweep kernel, so that an iteration reads its row block once (pg_gemv_tn4.hip)
struct pg_row_team {
  int n = 0, rank = 0;        // devices in the team (0 / 1: off), this device's index
  void* inbox[16] = {};       // every device's inbox as mapped HERE (inbox[rank] is this device's own)
  void* own = nullptr;        // ... allocated by pg_ctx_row_team_alloc (freed with the context)
  int max_wgs = 0;            // workgroups per device (0: one per compute unit); must be the same on every device
  unsigned epoch = 0, scal_epoch = 0;  // launch epochs of the granule tags: advance in step on every device
  unsigned long long ring_sig = 0;     // layout of the granule ring the last sweep used (workgroups, C, G, LAG, devices)
  double* f_local = nullptr;  // device, two doubles: [0] this device's 1/2 lam ||r_p||^2 between the finish kernel and the exchange; [1] spare; [2..18): the one-hot slots of pg_mat_row_team_agree
  unsigned long long* wait_stats = nullptr;  // device: { late waves, polls spent waiting, (latency injector:) ticks of slack left, steps counted } since pg_ctx_set_row_team (telemetry)
  long long sweeps = 0;       // row-team sweeps launched since pg_ctx_set_row_team
  unsigned gen = 0;           // bumped by every pg_ctx_set_row_team: what a matrix agreed on with one team does not carry over
  bool solo = false;          // tests / profiling (pg_ctx_test_team_fault kind 4): a team of ONE device is a team -- the sweep posts to and polls its own
                              // inbox, so the kernel can run ALONE under a profiler that serialises kernels (rocprofv3 --pmc)
  // pg_ctx_row_team_tune: the sweep's geometry per context, WITHOUT PG_TUNE (0 = the table's choice; the same on every device of the team)
  struct Tune {
    int C = 0, LAG = 0, LAGR = 0, PF = 0, WGS = 0, W = 0;
    int K1 = -1;            // -1: default (the one-wave sweep where one wave holds the column); 0: round 5's kernel
    int PAIR = -1;          // -1 / 0: one post per step; 1: one post per two steps (half the fabric transactions)
    int AHEAD = -1;         // -1 / 1: the poll of a step's totals is issued one step ahead of its use; 0: at the start of its own step
    long long SPIN = 0;     // bounded wait in polls (0: 2^21)
  } tune;
  // what the LAST row-team sweep ran with (pg_ctx_row_team_geometry)
  struct Geom {
    int W = 0, U = 0, C = 0, LAG = 0, LAGR = 0, PF = 0, WGS = 0, K1 = 0, PAIR = 0, AHEAD = 0, nteams = 0;
    long long SPIN = 0;
  } last;
};

namespace pgtn {
size_t peer_inbox_bytes();  // pg_gemv_tn4.hip: granule ring + scalar inbox of one device
}

struct pg_ctx {
  int device = 0;
  pg_row_team rteam;
  std::vector<void*> rteam_imported;  // peers' inboxes opened through IPC handles (closed with the context)
  hipStream_t stream = nullptr;
  hipDeviceProp_t prop{};
  int num_cu = 256;
  // reduction workspace
  double* red_partials = nullptr;  // [PG_RED_MAX_BLOCKS * PG_RED_MAX_NS]
  unsigned* red_counter = nullptr; // [8]
  double* hscal = nullptr;         // [PG_S_COUNT]  scalar block: mapped pinned host memory
  double* dscal = nullptr;         //               ... and its device-side address (kernels store here)
  // collective
  pg_allreduce_fn allreduce = nullptr;
  pg_allreduce_fn allreduce_begin = nullptr;  // asynchronous issue (overlaps with following kernels)
  pg_allreduce_wait_fn allreduce_wait = nullptr;
  void* allreduce_user = nullptr;
  pg_comm* comm = nullptr;  // native RCCL path (optional)
  // column sharding (pg_ctx_set_column_sharding): this rank holds a column block of A and the matching slices of all
  // n-vectors; m-vectors are replicated and what crosses ranks is A x (m elements) and four epilogue scalars
  int shard_cols = 0, shard_nranks = 1, shard_rank = 0;
  double* small_out = nullptr;       // result block of the single-workgroup solver (device address)
  double* small_out_host = nullptr;  //   ... mapped pinned host memory
  void* coop_ws = nullptr;           // workspace of the cooperative solver (barrier counter, partials)
  void* dr_ws = nullptr;             // third x buffer of pg_dr_run's two-blocks-in-flight loop
  size_t dr_ws_bytes = 0;
  hipEvent_t dr_ev[2] = {nullptr, nullptr};
  size_t coop_ws_bytes = 0;
  // long-column sweep (gemv_tnt_kernel, pg_gemv_tn2.hip): a timeout seen by the last scalar read-back, and the test hook
  // pg_ctx_test_team_fault(ctx, k) (tests only; proxgrad_hip_ext.h): the k-th team launch on this context (1-based) goes out with one workgroup
  // missing, so that one team times out
  bool team_timeout = false;
  int test_team_fault = 0;
  int test_team_fault_kind = 0;  // 0: one workgroup never starts (its team times out); 1: the launch is refused
  // pg_ctx_test_team_fault(ctx, ns, 2): latency injector of the row-team sweep -- a step's granules are accepted by their consumers
  // only `ns` nanoseconds after they were stored (gemv_tnt_kernel<..., DELAY>, pg_gemv_tnt.h; members must share a device's clock)
  unsigned test_team_delay_ticks = 0;  // 100 MHz ticks; 0: off
  bool test_team_delay_on = false;     // the DELAY instantiation is launched (also with 0 ticks: the injector's own cost)
  long team_launches = 0;
  bool team_plain_launch = false;  // PG_TN_TEAM_PLAIN = 1: plain instead of cooperative launch (A/B measurements)
  // Cooperative queues of different PROCESSES are not run side by side on one device: next to any process that holds one
  // (even an idle one) the cooperative team sweep runs at 0.45 of its rate, silently (profiles/r3_team_coop_vs_plain.md).
  // The first cooperative sweeps of a context over a matrix of >= 1 GiB are therefore timed with an event pair; two in a row
  // below COOP_SLOW_BYTES_PER_S are reported: one line on stderr and PG_FLAG_COOP_SLOW (the remedy, PG_TN_TEAM_PLAIN=1, must
  // be in place before the process's first cooperative launch: switching afterwards does not help, profiles/r4_coop_probe.md).
  hipEvent_t coop_probe[2] = {nullptr, nullptr};
  double coop_probe_bytes = 0;  // bytes of the probed launch (0: no probe in flight)
  int coop_probes_left = 6, coop_slow_in_a_row = 0;
  bool coop_slow = false;  // reported once through the iteration flags
  // stream capture (pg_ctx_capture_begin / _end): launches are recorded into a hipGraph instead of executed; scalar
  // read-backs are skipped (their host values are not meaningful until the graph has run)
  bool capturing = false;
  // event-pair kernel timing (pg_ctx_profile_*)
  bool profiling = false;
  uint32_t prof_mask = 0xFFFFFFFFu;  // kernel kinds that get an event pair while profiling (pg_ctx_profile_select)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events[PG_K_COUNT];
  std::vector<hipEvent_t> prof_pool;
};

// RAII bracket: records an event pair around a kernel launch when profiling is enabled
struct pg_prof_scope {
  pg_ctx* c;
  int kind;
  hipEvent_t start = nullptr, stop = nullptr;
  pg_prof_scope(pg_ctx* ctx, int k);
  ~pg_prof_scope();
};

struct pg_graph {
  pg_ctx* ctx = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};

struct pg_mat {
  pg_ctx* ctx = nullptr;
  int dtype = PG_F32;
  int64_t m = 0, n = 0, ld = 0;  // ld in elements, multiple of 1 KiB / sizeof(T)
  void* data = nullptr;          // [ld * n], padding rows are zero
  // workspace for y = A x partial sums (lazy)
  void* partials = nullptr;
  int64_t partials_slots = 0;
  int team_nrg = 0;        // row team: the LONGEST row block of the team in row groups, agreed once per matrix and team (pg_mat_row_team_agree)
  unsigned team_nrg_gen = 0;
  std::vector<void*> retired;  // outgrown partial-sum buffers of a row-team matrix, freed with the matrix (pg_gemv_tn4.hip)
  void* rpad = nullptr;  // [ld] zero-padded copy of a caller's m-vector (pg_mat_fused_tn)
  void* rpad2 = nullptr; // ... of the second instance's (pg_mat_fused_tn_pair)
  void* rpad3 = nullptr; // ... of the third's (pg_mat_fused_tn_trio)
  void* xch = nullptr;   // granule ring of the workgroup teams of the long-column sweep (gemv_tnt_kernel)
  size_t xch_bytes = 0;
  unsigned xch_epoch = 0;     // launch epoch of the ring (1 .. 255, the high byte of every granule tag: no memset per launch)
  long long xch_layout = -1;  // (nteams, team size, C) the ring was last used with; a change re-zeroes it
};

struct pg_ls {
  pg_ctx* ctx = nullptr;
  pg_mat* A = nullptr;
  const void* b = nullptr;  // device m-vector (borrowed)
  double lam = 1.0;
  void* r = nullptr;     // [ld] residual A x - b
  void* gbuf = nullptr;  // [n + 1] gradient ++ f, the all-reduce payload
  void* gchunks = nullptr;  // [nchunks * n] partial gradients when m needs several LDS chunks
  void* cbuf = nullptr;     // [ld + 4 * nranks] column sharding: the all-reduce payload [A x partial ; scalar slots]
  int64_t a_passes = 0;     // telemetry: full reads of A
  uint64_t r_gen = 0;       // bumped whenever r is rewritten (single-sweep iterations check their speculation)
};

// which way a registered collective is used
static inline bool pg_rteam_active(const pg_ctx* c) { return c->rteam.n > 1 || (c->rteam.n == 1 && c->rteam.solo); }
static inline bool pg_row_sharded(const pg_ctx* c) { return (c->allreduce != nullptr || c->allreduce_begin != nullptr) && !c->shard_cols; }
static inline bool pg_col_sharded(const pg_ctx* c) { return (c->allreduce != nullptr || c->allreduce_begin != nullptr) && c->shard_cols; }

static inline size_t pg_sizeof(int dtype) { return dtype == PG_F64 ? 8 : 4; }
static inline int64_t pg_round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// scalar read-back: copy `count` doubles starting at slot `first` to the pinned mirror and sync
pg_status pg_read_scalars(pg_ctx* ctx, int first, int count);

// ---------------------------------------------------------------------------------------------
// internal (device-pointer, asynchronous) building blocks used by the fused iterations
// ---------------------------------------------------------------------------------------------
// r = A x - b (b may be null), f -> dscal[PG_S_F]; local rows only (no collective)
pg_status pg_ls_residual_async(pg_ls* f, const void* x);
// full evaluation: local passes, then the all-reduce of [grad ; f] when sharded.  Leaves the gradient in
// *grad_ptr_out (either grad_out or f->gbuf) and f in dscal[PG_S_F].
pg_status pg_ls_vg_async(pg_ls* f, const void* x, void* grad_out);
pg_status pg_ls_value_async(pg_ls* f, const void* x);
// ONE sweep over A: g = lam A' r from the residual held in f->r (that of x), the epilogue for (x, g, gamma), v = z_new +
// beta (z_new - z_old), then f->r = A v - b and dscal[PG_S_F] = lam/2 ||A v - b||^2; epilogue scalars in
// dscal[PG_S_GZ .. PG_S_RESSQ].  Unsharded operators with <= 128 row groups only (pg_ls_fused_pass_supported).
pg_status pg_ls_fused_pass_async(pg_ls* f, const void* r_src /* null: f->r */, void* r_dst /* null: f->r */,
                                 double* f_dst /* device scalar; null: dscal[PG_S_F] */, const void* x, const void* z_old,
                                 double gamma, double beta, int g_kind, double g_p0, double g_p1, void* grad, void* y,
                                 void* z_new, void* res, void* v_next /* nullable */,
                                 const void* g_v0 = nullptr /* IndBox: per-element lo, hi (else the scalars g_p0, g_p1) */,
                                 const void* g_v1 = nullptr);
bool pg_ls_fused_pass_supported(const pg_ls* f);
// Row teams: the LONGEST row block of the team in row groups -> A->team_nrg, agreed once per matrix and team through the registered
// all-reduce (pg_gemv.hip); every device of the team must get here at the same point of its call sequence
pg_status pg_mat_row_team_agree(pg_ctx* c, pg_mat* A);
// g = lam A' r (+ all-reduce) from the residual currently held in f->r; f must already be in dscal[PG_S_F]
pg_status pg_ls_grad_stage_async(pg_ls* f, void* grad_out);
// r_out = a r1 + b r2 over m elements, dscal[PG_S_F] = f_scale ||r_out||^2, optional typed mirror of f
pg_status pg_residual_combo_async(pg_ctx* ctx, int dtype, int64_t m, void* r_out, double a, const void* r1, double b,
                                  const void* r2, double f_scale, void* f_typed, double* f_dst = nullptr);
// One cooperative launch at a time per process: two host threads (two contexts) inside hipLaunchCooperativeKernel at once left the
// runtime in a state that crashed the process at exit (ROCm 7.2; reproduced with two threads x three team sweeps -- one after the
// other, or under this lock: clean).  Held for the enqueue only, not for the kernel.  (pg_core.hip)
std::mutex& pg_coop_launch_mutex();

// row teams: *out = sum over the devices of *local (device order), PG_S_TEAMERR = any device's flag; one launch, no collective
pg_status pg_rteam_sum_scalar(pg_ctx* ctx, const double* local, double* out);
// column sharding: global sums / max of the four epilogue scalars in dscal[PG_S_GZ..PG_S_RESSQ] (one small all-reduce)
pg_status pg_ls_allreduce_epilogue_scalars(pg_ls* f);
pg_status pg_fb_epilogue_async(pg_ctx* ctx, int dtype, int64_t n, const void* x, const void* grad, double gamma,
                               int g_kind, double g_p0, double g_p1, void* y, void* z, void* res, const void* g_v0 = nullptr,
                               const void* g_v1 = nullptr);

#ifdef __HIPCC__
// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <typename T>
struct VecOf;
template <>
struct VecOf<float> {
  using type = f32x4;
  static constexpr int N = 4;
};
template <>
struct VecOf<double> {
  using type = f64x2;
  static constexpr int N = 2;
};

// max that PROPAGATES NaN, like Julia's `max` and therefore `norm(x, Inf)` (forward_backward.jl:126, fast_forward_backward.jl:151: the
// stopping rule): fmax drops a NaN operand, and an iteration that has diverged to NaN then shows norm(res, Inf) = 0 and "converges"
// (found by the option-drawing fuzzer of round 6: minimum_gamma above the stable step).  Every max REDUCTION uses this one.
__host__ __device__ __forceinline__ double pg_maxn(double a, double b) { return (a != a || b != b) ? (a + b) : fmax(a, b); }
__host__ __device__ __forceinline__ float pg_maxn(float a, float b) { return (a != a || b != b) ? (a + b) : fmaxf(a, b); }
__device__ __forceinline__ double pg_shfl_down(double v, int off) { return __shfl_down(v, off, 64); }
__device__ __forceinline__ float pg_shfl_down(float v, int off) { return __shfl_down(v, off, 64); }
__device__ __forceinline__ double pg_shfl_xor(double v, int m) { return __shfl_xor(v, m, 64); }
__device__ __forceinline__ float pg_shfl_xor(float v, int m) { return __shfl_xor(v, m, 64); }

// DPP lane moves (no LDS round trip) for 4- and 8-byte values
template <int CTRL>
__device__ __forceinline__ float pg_dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double pg_dpp_mov(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
// Threshold of the weighted 1-norm's prox, gamma * lam_j, ROUNDED on its own: ProximalOperators forms gl = gamma * lambda[i]
// and then compares / adds it, so the product must not be contracted into the following add (the scalar case gets its
// product from the host already rounded).
template <typename T>
__device__ __forceinline__ T pg_l1w_threshold(T gamma, T lam) {
#pragma clang fp contract(off)
  const T th = gamma * lam;
  return th;
}

__device__ __forceinline__ float pg_readlane(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ double pg_readlane(double v, int lane) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const int lo = __builtin_amdgcn_readlane((int)(unsigned)b, lane);
  const int hi = __builtin_amdgcn_readlane((int)(unsigned)(b >> 32), lane);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
// 16-lane row reduction, result in every lane of the row: quad_perm xor 1, xor 2, then row_half_mirror / row_mirror
// (after the quad steps every lane of a quad holds the quad total, so the mirrored partner contributes exactly the
// other quad / the other half-row).  Fixed order.
template <bool MAX, typename T>
__device__ __forceinline__ T pg_row_allreduce(T v) {
  T o = pg_dpp_mov<0xB1>(v);
  v = MAX ? pg_maxn(v, o) : (v + o);
  o = pg_dpp_mov<0x4E>(v);
  v = MAX ? pg_maxn(v, o) : (v + o);
  o = pg_dpp_mov<0x141>(v);
  v = MAX ? pg_maxn(v, o) : (v + o);
  o = pg_dpp_mov<0x140>(v);
  v = MAX ? pg_maxn(v, o) : (v + o);
  return v;
}
// whole-wave reduction, result (wave-uniform) in every lane: row reductions, then the four row totals travel through
// scalar registers (v_readlane) and are combined as (r0 + r1) + (r2 + r3).  Every lane of the wave must be active.
template <bool MAX, typename T>
__device__ __forceinline__ T pg_wave_allreduce(T v) {
  v = pg_row_allreduce<MAX, T>(v);
  const T r0 = pg_readlane(v, 0), r1 = pg_readlane(v, 16), r2 = pg_readlane(v, 32), r3 = pg_readlane(v, 48);
  const T a = MAX ? pg_maxn(r0, r1) : (r0 + r1), b = MAX ? pg_maxn(r2, r3) : (r2 + r3);
  return MAX ? pg_maxn(a, b) : (a + b);
}

// Deterministic grid-wide reduction of NS doubles per thread (bit k of MAXMASK: slot k is a max, else a
// sum).  NW-wave (NW*64-thread) blocks, every thread of the block must call it; gridDim.x <= PG_RED_MAX_BLOCKS.  Every block publishes its partial with
// write-through agent-scope stores; the last block to arrive (ticket counter) combines all partials in a
// fixed order and writes out[k] * post_scale[k].  The counter is reset for the next launch.  Returns true
// (to all its threads) in the finalizing block only; out[] is then visible to that block's thread 0.
// WAVE_REDUCED: v[] already holds the wave's values in lane 0 (the caller reduced within the wave, e.g. in working precision
// with DPP moves); the fp64 shuffle stage is skipped.
// slot k is a max when bit k of the mask is set; slots from 64 on are sums (the 64-iteration Douglas-Rachford block: 64 maxima + 2 sums)
__host__ __device__ constexpr bool pg_red_is_max(unsigned long long mask, int k) { return k < 64 && ((mask >> k) & 1ull) != 0; }

template <int NS, unsigned long long MAXMASK, int NW = 4, bool WAVE_REDUCED = false>
__device__ __forceinline__ bool grid_reduce_finalize(double (&v)[NS], double* __restrict__ partials,
                                                     unsigned* __restrict__ counter, double* __restrict__ out,
                                                     const double (&post_scale)[NS], double* final_vals = nullptr) {
  __shared__ double sm[NW * NS];
  __shared__ int sm_last;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  if constexpr (!WAVE_REDUCED) {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        double o = pg_shfl_down(v[k], off);
        v[k] = pg_red_is_max(MAXMASK, k) ? pg_maxn(v[k], o) : (v[k] + o);
      }
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NS; ++k) sm[wave * NS + k] = v[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // Publish this workgroup's partial with 8-byte agent-scope (write-through, sc1) stores, drain them, then
    // take a ticket.  No release/acquire fences: a release here would write back every dirty line the kernel
    // body left in this XCD's L2 (measured: several us per launch on the streaming kernels); the partials are
    // read back below with agent-scope loads, which bypass the reader's L1.
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      double a = sm[k];
      for (int w = 1; w < NW; ++w) a = pg_red_is_max(MAXMASK, k) ? pg_maxn(a, sm[w * NS + k]) : (a + sm[w * NS + k]);
      __hip_atomic_store(&partials[(size_t)blockIdx.x * NS + k], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sm_last = (t == gridDim.x - 1);
  }
  __syncthreads();
  if (!sm_last) return false;
  double acc[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) acc[k] = pg_red_is_max(MAXMASK, k) ? -INFINITY : 0.0;
  for (unsigned b = threadIdx.x; b < gridDim.x; b += NW * 64) {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      double p = __hip_atomic_load(&partials[(size_t)b * NS + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc[k] = pg_red_is_max(MAXMASK, k) ? pg_maxn(acc[k], p) : (acc[k] + p);
    }
  }
#pragma unroll
  for (int k = 0; k < NS; ++k) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      double o = pg_shfl_down(acc[k], off);
      acc[k] = pg_red_is_max(MAXMASK, k) ? pg_maxn(acc[k], o) : (acc[k] + o);
    }
  }
  __syncthreads();  // sm reuse
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NS; ++k) sm[wave * NS + k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      double a = sm[k];
      for (int w = 1; w < NW; ++w) a = pg_red_is_max(MAXMASK, k) ? pg_maxn(a, sm[w * NS + k]) : (a + sm[w * NS + k]);
      out[k] = a * post_scale[k];
      if (final_vals != nullptr) final_vals[k] = a * post_scale[k];  // valid on thread 0 of the finalizing block
    }
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return true;  // every thread of the finalizing block
}

// The same protocol for MANY slots (the 32- and 64-iteration Douglas-Rachford blocks: K maxima + 2 sums) without holding
// NS doubles per thread: grid_reduce_finalize keeps v[NS] and acc[NS] live (4 NS registers -- 270 at NS = 66, one wave per
// SIMD for the whole kernel).  Here the caller hands over a functor that produces slot k's WAVE-reduced value (valid in lane
// 0) on demand, and the finalizing block combines the partials eight slots at a time.  Same layout of `partials`, same ticket.
template <int NS, unsigned long long MAXMASK, int NW, typename WaveVal>
__device__ __forceinline__ bool grid_reduce_finalize_streamed(WaveVal wave_val, double* __restrict__ partials,
                                                              unsigned* __restrict__ counter, double* __restrict__ out,
                                                              double scale_last) {
  constexpr int CH = 8;
  __shared__ double sm[NW * NS];
  __shared__ int sm_last;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const double v = wave_val(k);
    if (lane == 0) sm[wave * NS + k] = v;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < NS; k += NW * 64) {  // one slot per thread: publish this workgroup's partial
    const bool is_max = pg_red_is_max(MAXMASK, k);
    double a = sm[k];
    for (int w = 1; w < NW; ++w) a = is_max ? pg_maxn(a, sm[w * NS + k]) : (a + sm[w * NS + k]);
    __hip_atomic_store(&partials[(size_t)blockIdx.x * NS + k], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // every publishing thread has drained its stores before the ticket is taken
  if (threadIdx.x == 0) {
    unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sm_last = (t == gridDim.x - 1);
  }
  __syncthreads();
  if (!sm_last) return false;
  for (int k0 = 0; k0 < NS; k0 += CH) {
    double acc[CH];
#pragma unroll
    for (int q = 0; q < CH; ++q) acc[q] = pg_red_is_max(MAXMASK, k0 + q) ? -INFINITY : 0.0;
    for (unsigned b = threadIdx.x; b < gridDim.x; b += NW * 64) {
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        if (k0 + q < NS) {
          const double p = __hip_atomic_load(&partials[(size_t)b * NS + k0 + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          acc[q] = pg_red_is_max(MAXMASK, k0 + q) ? pg_maxn(acc[q], p) : (acc[q] + p);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < CH; ++q) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const double o = pg_shfl_down(acc[q], off);
        acc[q] = pg_red_is_max(MAXMASK, k0 + q) ? pg_maxn(acc[q], o) : (acc[q] + o);
      }
    }
    __syncthreads();  // sm reuse
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < CH; ++q) sm[wave * CH + q] = acc[q];
    }
    __syncthreads();
    if (threadIdx.x < CH && k0 + (int)threadIdx.x < NS) {
      const int k = k0 + threadIdx.x;
      const bool is_max = pg_red_is_max(MAXMASK, k);
      double a = sm[threadIdx.x];
      for (int w = 1; w < NW; ++w) a = is_max ? pg_maxn(a, sm[w * CH + threadIdx.x]) : (a + sm[w * CH + threadIdx.x]);
      out[k] = k == NS - 1 ? a * scale_last : a;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}
// Tuning knobs (PG_N_*, PG_T_*, PG_TN*, ...) are for experiments: they are looked up only in a process that was started
// with PG_TUNE set (read once); a production process never calls getenv on the launch path and cannot be steered by a
// stray variable.
inline bool pg_tuning_enabled() {
  static const bool on = getenv("PG_TUNE") != nullptr;
  return on;
}
inline const char* env_str(const char* name) { return pg_tuning_enabled() ? getenv(name) : nullptr; }

#endif  // __HIPCC__
