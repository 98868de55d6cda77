// Row teams: north_star's ROW layout (GPU p holds the row block A_p, the n-vectors are replicated) at ONE read of A per
// iteration.
//
// benchmark/benchmarks.jl:15-16 are two products: res = A x - b, then A' res.  Under row blocks A' res = sum_p A_p' res_p is
// needed globally before the prox, so the row-sharded iteration reads its block twice: sweep, all-reduce of
// n + 1 elements, replicated epilogue, sweep.  Here the devices form PEER teams instead (gemv_tnt_kernel<..., PEER>,
// pg_gemv_tnt.h): workgroup w of every device walks the same column groups in the same order; for a column j device p forms
// its partial dot A_p[:, j]' res_p and pushes it, as one tagged 8-byte granule, into the inbox of every device; LAG steps
// later every device finds all partials in its OWN memory, sums them in device order (the same bits everywhere), applies
// the prox and accumulates A_p[:, j] v_j from the tile that waited in LDS.  What crosses the fabric per iteration is
// 8 (N - 1) n bytes of granule pushes per device (58 MB at N = 8, n = 2^20) instead of a 4 MiB all-reduce -- more bytes, but
// no second read of the 8 GiB block.  After the sweep: r_p = A_p v - b_p and 1/2 ||r_p||^2 locally (gemv_n_finish), then
// peer_scalars_kernel exchanges { 1/2 ||r_p||^2, this device's timeout flag } the same way, so that f and the decision to
// fall back are the same on every device without a collective.
//
// Reference statements: benchmark/benchmarks.jl:15-16, fast_forward_backward.jl:135-142 (forward_backward.jl:113-120).
#include <mutex>
#include <type_traits>

#include "pg_gemv_tn.h"

namespace pgtn {
pg_status peer_scalar_exchange(pg_ctx* c, const double* f_local, double* f_out);
namespace {
#include "pg_gemv_tnt.h"
#include "pg_gemv_tnp1.h"

constexpr int PEER_TEAMS_MAX = 1024;  // workgroups per device the inbox has ring space for (up to four per compute unit)
constexpr size_t PEER_RING_BYTES = (size_t)PEER_TEAMS_MAX * PEER_RING * (size_t)(TEAM_MAX * 8) * sizeof(unsigned long long);  // C * G <= 8
constexpr int PEER_SCAL_GRANULES = 4;  // per device and slot: f (two halves), the timeout flag, one spare
constexpr size_t PEER_SCAL_BYTES = 2 * (size_t)TEAM_MAX * PEER_SCAL_GRANULES * sizeof(unsigned long long);

struct PeerScalArgs {
  int n, rank;
  unsigned tag;
  int slot;
  unsigned long long* inbox[TEAM_MAX];  // every device's scalar inbox as seen from here
  const double* f_local;                // this device's 1/2 lam ||r_p||^2
  double* f_out;                        // sum over the devices, in device order
  long long spin_limit;                 // polls before the wave gives up
  unsigned aux;                         // a small integer every device contributes (the spare granule) ...
  double* aux_max_out;                  // ... and where the largest of them goes (nullptr: nobody asked; 0 when a peer never answered)
  double* team_err;                     // in: this device's flag; out: any device's
};

// one wave: post {f lo, f hi, err} into every inbox, wait for all of them here, combine in device order
__global__ __launch_bounds__(64) void peer_scalars_kernel(PeerScalArgs p) {
  const int lane = threadIdx.x;
  const double f = *p.f_local;
  const unsigned long long fb = __builtin_bit_cast(unsigned long long, f);
  const unsigned err = *p.team_err != 0.0 ? 1u : 0u;
  const size_t base = (size_t)p.slot * TEAM_MAX * PEER_SCAL_GRANULES;
  if (lane < 4) {
    const unsigned bits = lane == 0 ? (unsigned)fb : (lane == 1 ? (unsigned)(fb >> 32) : (lane == 2 ? err : p.aux));
    const unsigned long long word = ((unsigned long long)p.tag << 32) | bits;
    for (int q = 0; q < p.n; ++q)
      __hip_atomic_store(p.inbox[q] + base + (size_t)p.rank * PEER_SCAL_GRANULES + lane, word, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const int npoll = p.n * PEER_SCAL_GRANULES;
  const int pl = lane < npoll ? lane : npoll - 1;
  const unsigned long long* src = p.inbox[p.rank] + base + pl;
  unsigned long long w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  bool dead = false;
  long long spins = 0;
  while (__builtin_amdgcn_ballot_w64((unsigned)(w >> 32) == p.tag) != ~0ull) {
    __builtin_amdgcn_s_sleep(2);
    w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (++spins > p.spin_limit) {
      dead = true;
      break;
    }
  }
  const int lo = (int)(unsigned)w;
  double total = 0.0;
  unsigned any = dead ? 1u : 0u, aux_max = 0u;
  for (int q = 0; q < p.n; ++q) {
    const unsigned l0 = (unsigned)__shfl(lo, q * PEER_SCAL_GRANULES + 0);
    const unsigned l1 = (unsigned)__shfl(lo, q * PEER_SCAL_GRANULES + 1);
    const unsigned l2 = (unsigned)__shfl(lo, q * PEER_SCAL_GRANULES + 2);
    const unsigned l3 = (unsigned)__shfl(lo, q * PEER_SCAL_GRANULES + 3);
    total += __builtin_bit_cast(double, ((unsigned long long)l1 << 32) | l0);
    any |= l2;
    aux_max = l3 > aux_max ? l3 : aux_max;
  }
  if (lane == 0) {
    *p.f_out = total;
    if (any) *p.team_err = 1.0;
    if (p.aux_max_out != nullptr) *p.aux_max_out = dead ? 0.0 : (double)aux_max;
  }
}

// ensure_partials without its hipFree: a free synchronises the whole DEVICE, and where several members of a team share one
// (the one-GPU tests: contexts of one process) a peer's kernel may already be waiting there for this rank's granules -- the
// free would sit behind that wait until it expires (measured: the first sweep of about one solve in ten fell back to two
// reads that way).  The outgrown buffer is kept until the matrix goes (at most a few per matrix: the buffer only grows).
pg_status grow_partials_without_free(pg_mat* A, int S) {
  if (A->partials && A->partials_slots >= S) return PG_OK;
  void* grown = nullptr;
  const size_t bytes = (size_t)S * (size_t)A->ld * pg_sizeof(A->dtype);
  hipError_t e = hipMalloc(&grown, bytes);
  if (e != hipSuccess) {
    pg_set_error("hipMalloc(%zu) for GEMV partial sums failed: %s", bytes, hipGetErrorString(e));
    return PG_ERR_ALLOC;
  }
  if (A->partials) A->retired.push_back(A->partials);  // kernels enqueued earlier may still read it
  A->partials = grown;
  A->partials_slots = S;
  return PG_OK;
}

// K1: gemv_tnp1_kernel (pg_gemv_tnp1.h, one wave per column: WAVES = 1) instead of gemv_tnt_kernel<..., PEER>; same protocol
template <typename T, int U, int C, int LAG, int PF, int LAGR = 0, bool DELAY = false, int WAVES = 4, bool K1 = false, bool PAIR = false, bool AHEAD = false>
pg_status launch_tnp(pg_mat* A, TNArgs<T>& a, int* blocks_out, int wgs_per_cu) {
  constexpr int G = (int)sizeof(T) / 4;
  constexpr int MS = (PAIR ? 2 : 1) * C * G;  // granules of one member in a ring slot
  static_assert(MS <= 8, "the inbox holds eight granules per member and ring slot");
  static_assert(!PAIR || K1, "one post per two steps: the one-wave sweep only");
  pg_ctx* c = A->ctx;
  const pg_row_team& rt = c->rteam;
  const int64_t ncg = (A->n + C - 1) / C;
  if (rt.n * MS > 64) {
    pg_set_error("a row team of %d devices with %d granules per device and ring slot needs more than one lane per granule", rt.n, MS);
    return PG_ERR_UNSUPPORTED;
  }
  if (DELAY && (rt.n * (MS + 1) > 64 || rt.n * (MS + 1) > TEAM_MAX * MS)) {
    pg_set_error("the latency injector's stamps do not fit beside the granules of %d devices", rt.n);
    return PG_ERR_UNSUPPORTED;
  }
  // as many workgroups per compute unit as the parked tiles leave room for (four of 32 KiB, two of 64 KiB, one of 128 KiB):
  // measured on 2048- / 4096- / 8192-row blocks, two members sharing one device, 5.87 (four) / 6.08 (two) / 6.28 (two) TB/s
  // against 3.31 / 5.16 / 6.19 with one (profiles/r4_row_team_one_gpu.md)
  constexpr size_t PARK = (size_t)LAG * WAVES * C * U * 1024;
  // (exact-U geometries park 40 / 48 KiB: three of them fit beside the kernel's static LDS).  Tiles that wait in registers
  // (LAGR) count against the register file instead: the caller's table says how many workgroups of this instantiation a
  // compute unit holds (wgs_per_cu; it must be the same number on every device -- same binary, same table).
  int64_t nteams = (int64_t)c->num_cu * (wgs_per_cu > 0 ? wgs_per_cu : (PARK <= 32 * 1024 ? 4 : PARK <= 48 * 1024 ? 3 : PARK <= 64 * 1024 ? 2 : 1));
  if (rt.max_wgs > 0) nteams = rt.max_wgs;
  if (rt.max_wgs < 0) nteams = nteams / -rt.max_wgs > 0 ? nteams / -rt.max_wgs : 1;  // -k: this device is shared by k members of the team
  if (nteams * MS > (int64_t)PEER_TEAMS_MAX * 8) nteams = (int64_t)PEER_TEAMS_MAX * 8 / MS;  // the inbox's ring space: nteams x RING x TEAM_MAX x MS granules
  if (nteams > PG_RED_MAX_BLOCKS) nteams = PG_RED_MAX_BLOCKS;
  if (nteams > ncg) nteams = ncg;
  if (nteams < 1) nteams = 1;
  PG_TRY(grow_partials_without_free(A, (int)nteams));
  a.partials = (T*)A->partials;
  a.team_size = 1;
  a.ueff = (a.nrg + WAVES - 1) / WAVES;
  a.deal_even = 1;
  if (a.ueff > U) {
    pg_set_error("gemv_tnt<U = %d, PEER> launched for %d row groups per wave", U, a.ueff);
    return PG_ERR_INVALID;
  }
  const int64_t steps = (ncg + nteams - 1) / nteams;
  if (steps + 1 >= (1 << 24)) {
    pg_set_error("the row-team sweep covers at most 2^24 column groups per workgroup (%lld here)", (long long)steps);
    return PG_ERR_UNSUPPORTED;
  }
  a.nteams = (int)nteams;
  a.peer_n = rt.n;
  a.peer_rank = rt.rank;
  for (int q = 0; q < rt.n; ++q) a.peer_ring[q] = (unsigned long long*)rt.inbox[q];
  a.xch = a.peer_ring[rt.rank];
  a.team_err = c->dscal + PG_S_TEAMERR;
  a.wait_stats = c->rteam.wait_stats;
  a.delay_ticks = c->test_team_delay_ticks;
  if (c->rteam.tune.SPIN > 0) a.spin_limit = c->rteam.tune.SPIN;
#ifdef PG_TNT_EXPERIMENT
  a.dbg = env_int("PG_TNT_DBG", 0);
  if (a.dbg & 2048) a.delay_ticks = (unsigned)env_int("PG_TNT_PACE", 250);
  if (a.dbg & 8192) a.delay_ticks = (unsigned)env_int("PG_TNT_FAKE", 8);
  if (K1) a.line_cols = env_int("PG_TNT_LINE_COLS", a.line_cols);  // (experiment: chunks of 16 / 64 columns -- twice / half the output store instructions)
#endif
  c->rteam.sweeps++;
  // The tags make a slot self-describing only among launches of ONE ring layout (every launch rewrites every slot it polls, so
  // a granule of the same epoch 254 launches ago is long gone).  When the layout changes -- another matrix shape, another
  // geometry -- an address may still hold a granule of the old layout's launch with the very tag the new one will wait for.
  // So on a layout change every device clears its own inbox and the devices meet in one scalar exchange before the sweep: a
  // peer starts pushing granules of the new layout only after that exchange, i.e. after this device's clear (stream order);
  // and nothing of the previous launch is still in flight, because ITS scalar exchange has completed everywhere.  All devices
  // see the change at the same launch (same sequence of calls), so the extra exchange pairs up.
  const unsigned long long sig = ((unsigned long long)nteams << 32) | ((unsigned long long)C << 24) | ((unsigned long long)G << 16) |
                                 ((unsigned long long)(LAG + LAGR) << 8) | ((unsigned long long)(DELAY ? 1 : 0) << 15) | ((unsigned long long)(PAIR ? 1 : 0) << 23) |
                                 (unsigned long long)rt.n;
  if (sig != c->rteam.ring_sig) {
    PG_HIP(hipMemsetAsync(rt.inbox[rt.rank], 0, PEER_RING_BYTES, c->stream));
    PG_TRY(peer_scalar_exchange(c, c->rteam.f_local, c->rteam.f_local));
    c->rteam.ring_sig = sig;
  }
  // every device launches the same sweeps in the same order, so the epochs agree without being communicated
  c->rteam.epoch = (c->rteam.epoch % 254u) + 1u;
  a.tag_base = c->rteam.epoch << 24;
  *blocks_out = (int)nteams;
  const size_t lds = (size_t)LAG * WAVES * C * U * 1024;
  static_assert(!K1 || WAVES == 1, "gemv_tnp1_kernel is the one-wave sweep");
  const void* kern;
  if constexpr (K1) kern = reinterpret_cast<const void*>(&gemv_tnp1_kernel<T, U, C, LAG, PF, LAGR, DELAY, PAIR, AHEAD>);
  else kern = reinterpret_cast<const void*>(&gemv_tnt_kernel<T, U, C, WAVES, LAG, PF, true, LAGR, DELAY, AHEAD>);
  if (lds + 4096 > 64 * 1024) {
    static std::mutex mu;
    static bool opted_in[64] = {};
    std::lock_guard<std::mutex> lock(mu);
    const int dev = c->device & 63;
    if (!opted_in[dev]) {
      PG_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      opted_in[dev] = true;
    }
  }
  unsigned grid = (unsigned)nteams;
  c->team_launches++;
  if (c->test_team_fault > 0 && c->team_launches == c->test_team_fault && grid > 1) {
    // test hook: one workgroup of this device never starts; its peers on the other devices time out.  (Both kinds: "refused at
    // launch" is a property of COOPERATIVE launches -- the one-device team sweep -- and this is a plain one; a device that
    // dropped out of a step on its own would leave its peers alone in the step's exchanges.)
    grid -= 1;
  }
  pg_prof_scope prof(c, PG_K_GEMV_TN);
  // a plain launch: co-residency across devices is nobody's promise, the members' waits are bounded instead
  c->rteam.last = {WAVES, U, C, LAG, LAGR, PF, wgs_per_cu, K1 ? 1 : 0, PAIR ? 1 : 0, AHEAD ? 1 : 0, (int)nteams, a.spin_limit};
#ifdef PG_TNT_EXPERIMENT
  const unsigned k1_block = (a.dbg & 2048) ? 128u : 64u;
#else
  const unsigned k1_block = 64u;
#endif
  if constexpr (K1) hipLaunchKernelGGL((gemv_tnp1_kernel<T, U, C, LAG, PF, LAGR, DELAY, PAIR, AHEAD>), dim3(grid), dim3(k1_block), lds, c->stream, a);
  else hipLaunchKernelGGL((gemv_tnt_kernel<T, U, C, WAVES, LAG, PF, true, LAGR, DELAY, AHEAD>), dim3(grid), dim3(WAVES * 64), lds, c->stream, a);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

}  // namespace

#ifndef PG_TN4_FLOAT64_UNIT
bool tn_peer_covers(int nrg) { return nrg >= 1 && nrg <= 64; }
#endif

namespace {
struct PeerGeom {
  int W, C, LAG, LAGR, PF, WGS;
};
// Geometry of the row-team sweep by the team's longest block in row groups of 1 KiB (profiles/r5_row_team_latency_sweep.md; measured
// in Float32 -- Float64 takes the same table by row groups: 128 rows each, the same registers per row group, two granules per value).
// One rule for every length: a WAVE streams 16 KiB tiles of its rows (U <= 8 row groups of C = 2 columns, or U <= 16 of one
// column), four waves per compute unit (one per SIMD, the whole 512-entry register file each), so W = ceil(row groups / 16)
// waves share a column -- ONE up to 2048 rows: no barrier, no cross-wave sum, the wave posts and polls for itself -- and 4 / W
// workgroups sit on a compute unit; two tiles in flight (PF), two waiting in LDS (LAG: 128 KiB per compute unit) and two more
// in registers (LAGR): 256 KiB of parked tiles per compute unit, 10-11 us for a step's granules to arrive where round 4's
// geometries (four waves per column at every length, LAG = 2 only) had 5-6.
PeerGeom peer_geometry(int nrg, bool f64) {
  const int W = nrg <= 8 ? 1 : nrg <= 32 ? 2 : 4;
  const int U = (nrg + W - 1) / W;
  // (Float64: C = 2 also for the shortest blocks -- 16 devices x C columns x two granules must fit the 64 lanes that poll them)
  return {W, U <= 4 && !f64 ? 4 : U <= 8 ? 2 : 1, 2, 2, 2, 4 / W};
}
}  // namespace

// every instantiation (both element types): (U, C, LAG, PF, LAGR, W); _D: also with the latency injector (the geometries of the latency sweep)
#define PG_TNP_GEOMETRIES \
  PG_TNP_CASE_D(8, 2, 2, 2, 2, 1); PG_TNP_CASE_D(8, 2, 2, 2, 2, 2); PG_TNP_CASE_D(16, 1, 2, 2, 2, 2); \
  PG_TNP_CASE_D(16, 1, 2, 2, 2, 4); PG_TNP_CASE_D(2, 2, 2, 2, 0, 4); PG_TNP_CASE_D(4, 2, 2, 2, 0, 4); \
  PG_TNP_CASE_D(8, 1, 2, 2, 0, 4); PG_TNP_CASE_D(16, 1, 2, 2, 0, 4); PG_TNP_CASE_D(16, 1, 2, 2, 1, 4); \
  PG_TNP_CASE(5, 2, 2, 2, 2, 2); \
  PG_TNP_CASE(6, 2, 2, 2, 2, 2); PG_TNP_CASE(7, 2, 2, 2, 2, 2); PG_TNP_CASE(9, 1, 2, 2, 2, 2); PG_TNP_CASE(10, 1, 2, 2, 2, 2); \
  PG_TNP_CASE(11, 1, 2, 2, 2, 2); PG_TNP_CASE(12, 1, 2, 2, 2, 2); PG_TNP_CASE(13, 1, 2, 2, 2, 2); PG_TNP_CASE(14, 1, 2, 2, 2, 2); \
  PG_TNP_CASE(15, 1, 2, 2, 2, 2); PG_TNP_CASE(9, 1, 2, 2, 2, 4); PG_TNP_CASE(10, 1, 2, 2, 2, 4); PG_TNP_CASE(11, 1, 2, 2, 2, 4); \
  PG_TNP_CASE(12, 1, 2, 2, 2, 4); PG_TNP_CASE(13, 1, 2, 2, 2, 4); PG_TNP_CASE(14, 1, 2, 2, 2, 4); PG_TNP_CASE(15, 1, 2, 2, 2, 4)
// ... of gemv_tnp1_kernel, the one-wave sweep of blocks up to 8 row groups (round 6; W = 1): (U, C, LAG, PF, LAGR)
#define PG_TNP1_GEOMETRIES_F32 \
  PG_TNP1_CASE(1, 4, 2, 2, 2); PG_TNP1_CASE(2, 4, 2, 2, 2); PG_TNP1_CASE(3, 4, 2, 2, 2); PG_TNP1_CASE(4, 4, 2, 2, 2); \
  PG_TNP1_CASE(5, 2, 2, 2, 2); PG_TNP1_CASE(6, 2, 2, 2, 2); PG_TNP1_CASE(7, 2, 2, 2, 2); PG_TNP1_CASE_D(8, 2, 2, 2, 2); PG_TNP1_DENSE
#ifdef PG_TNT_EXPERIMENT
#define PG_TNP1_DENSE PG_TNP1_CASE(8, 1, 2, 2, 1); PG_TNP1_CASE(8, 1, 2, 2, 2); PG_TNP1_CASE(8, 1, 3, 2, 1)
#else
#define PG_TNP1_DENSE
#endif
#define PG_TNP1_GEOMETRIES_F64 \
  PG_TNP1_CASE(1, 2, 2, 2, 2); PG_TNP1_CASE(2, 2, 2, 2, 2); PG_TNP1_CASE(3, 2, 2, 2, 2); PG_TNP1_CASE(4, 2, 2, 2, 2); \
  PG_TNP1_CASE(5, 2, 2, 2, 2); PG_TNP1_CASE(6, 2, 2, 2, 2); PG_TNP1_CASE(7, 2, 2, 2, 2); PG_TNP1_CASE_D(8, 2, 2, 2, 2)

// Tunables (environment, under PG_TUNE, for experiments): PG_TNP_W, PG_TNP_C, PG_TNP_LAG, PG_TNP_LAGR, PG_TNP_PF, PG_TNP_WGS.
template <typename T>
pg_status launch_tn_peer(pg_mat* A, TNArgs<T>& a, int* blocks_out) {
  pg_ctx* c = A->ctx;
  // the longest row block of the team, agreed once per matrix and team (pg_gemv.hip::pg_mat_row_team_agree: a collective, normally
  // made when the iterator was created); a block beyond what the sweep covers on ANY device: no device sweeps
  PG_TRY(pg_mat_row_team_agree(c, A));
  if (!tn_peer_covers(A->team_nrg) || A->team_nrg < a.nrg) {
    pg_set_error("the longest row block of the team (%d row groups) is beyond the row-team sweep's %d", A->team_nrg, 64);
    return PG_ERR_UNSUPPORTED;
  }
  const int team_nrg = A->team_nrg;
  // Geometry by the team's longest block (peer_geometry): waves per workgroup W (the rows of a column are split over them), row
  // groups per wave U = ceil(row groups / W) EXACTLY (a block one row past a boundary is one more row group per wave, not the next
  // power of two's geometry: 2 x 2049 rows streamed 4.0 TB/s that way in round 4 where 2 x 2048 streamed 5.9; PG_TNP_EXACT=0
  // under PG_TUNE: the powers of two only, for A/B runs), columns per step C, lag steps in LDS (LAG) and in registers (LAGR),
  // tiles in flight (PF), workgroups per compute unit (0: as many as the parked tiles leave LDS for).
  const PeerGeom g = peer_geometry(team_nrg, sizeof(T) == 8);
  // every knob: the context's own setting (pg_ctx_row_team_tune -- no PG_TUNE needed, what a first run on real fabric turns), else
  // the tuning variable (PG_TUNE processes only), else the table
  const pg_row_team::Tune& tn = c->rteam.tune;
  auto knob = [](int set, const char* var, int table) { return set > 0 ? set : env_int(var, table); };
  const int W = knob(tn.W, "PG_TNP_W", g.W);
  const int per_wave = (team_nrg + W - 1) / W;
  int U = per_wave < 2 && W == 4 ? 2 : per_wave;
  if (env_int("PG_TNP_EXACT", 1) == 0) {
    U = 2;
    while (U < per_wave) U *= 2;
  }
  const int C = knob(tn.C, "PG_TNP_C", g.C), LAG = knob(tn.LAG, "PG_TNP_LAG", g.LAG);
  const int LAGR = tn.LAGR > 0 ? tn.LAGR - 1 : env_int("PG_TNP_LAGR", g.LAGR);  // (0 is a value here: the knob carries LAGR + 1)
  const int PF = knob(tn.PF, "PG_TNP_PF", g.PF), WGS = knob(tn.WGS, "PG_TNP_WGS", g.WGS);
  const bool delay = c->test_team_delay_on;
  // one wave per column (W = 1): gemv_tnp1_kernel; K1 = 0: round 5's gemv_tnt_kernel<..., W = 1> (kept at U = 8 for the A/B)
  const bool k1 = W == 1 && (tn.K1 >= 0 ? tn.K1 != 0 : env_int("PG_TNP_K1", 1) != 0);
  // one post per two steps, where the pair's granules of all devices still fit the 64 polling lanes (else: one post per step)
  const bool pair_req = tn.PAIR >= 0 ? tn.PAIR != 0 : env_int("PG_TNP_PAIR", 0) != 0;
  // the poll of a step's totals one step ahead of its use (pg_gemv_tnp1.h): default; AHEAD = 2 (tune) / PG_TNP_AHEAD=0: at the start of its own step
  const bool ahead = tn.AHEAD >= 0 ? tn.AHEAD != 0 : env_int("PG_TNP_AHEAD", 1) != 0;
  const int ms2 = 2 * C * (int)(sizeof(T) / 4);  // granules of one device in a pair's ring slot
  const bool pair = pair_req && k1 && ms2 <= 8 && c->rteam.n * (ms2 + (delay ? 1 : 0)) <= 64;
#define PG_TNP1_ONE(UU, CC, LL, PP, RR, DD, PR, AH) \
  if (k1 && pair == PR && ahead == AH && U == UU && C == CC && LAG == LL && PF == PP && LAGR == RR && delay == DD) return launch_tnp<T, UU, CC, LL, PP, RR, DD, 1, true, PR, AH>(A, a, blocks_out, WGS)
#define PG_TNP1_CASE(UU, CC, LL, PP, RR) \
  PG_TNP1_ONE(UU, CC, LL, PP, RR, false, false, true); PG_TNP1_ONE(UU, CC, LL, PP, RR, false, true, true)
  // _D: also with the latency injector, and with the poll at the start of its own step (AHEAD = 0: round 6's first form, for the A/B)
#define PG_TNP1_CASE_D(UU, CC, LL, PP, RR) \
  PG_TNP1_CASE(UU, CC, LL, PP, RR); PG_TNP1_ONE(UU, CC, LL, PP, RR, true, false, true); PG_TNP1_ONE(UU, CC, LL, PP, RR, true, true, true); \
  PG_TNP1_ONE(UU, CC, LL, PP, RR, false, false, false); PG_TNP1_ONE(UU, CC, LL, PP, RR, true, false, false)
  if constexpr (sizeof(T) == 8) {
    PG_TNP1_GEOMETRIES_F64;
  } else {
    PG_TNP1_GEOMETRIES_F32;
  }
#undef PG_TNP1_CASE
#undef PG_TNP1_CASE_D
#undef PG_TNP1_ONE
#define PG_TNP_ONE(UU, CC, LL, PP, RR, WW, DD, AH) \
  if (!k1 && ahead_w == AH && U == UU && C == CC && LAG == LL && PF == PP && LAGR == RR && W == WW && delay == DD) return launch_tnp<T, UU, CC, LL, PP, RR, DD, WW, false, false, AH>(A, a, blocks_out, WGS)
#define PG_TNP_CASE(UU, CC, LL, PP, RR, WW) PG_TNP_ONE(UU, CC, LL, PP, RR, WW, false, false)
#define PG_TNP_CASE_D(UU, CC, LL, PP, RR, WW) PG_TNP_CASE(UU, CC, LL, PP, RR, WW); PG_TNP_ONE(UU, CC, LL, PP, RR, WW, true, false)
  // ... with the poll one step ahead (LT = 4 geometries of several waves per column; A/B: PG_TNP_AHEAD)
#define PG_TNP_CASE_A(UU, CC, LL, PP, RR, WW) PG_TNP_ONE(UU, CC, LL, PP, RR, WW, false, true); PG_TNP_ONE(UU, CC, LL, PP, RR, WW, true, true)
  // (measured, two ranks on one device: +1.5 % at 2 x 4096 rows, +1.3 % at 2 x 16384, where U = 16 spills four registers -- so only on request:
  // pg_ctx_row_team_tune(ctx, "AHEAD", 1) or PG_TNP_AHEAD_W=1 under PG_TUNE; the one-wave sweep has it by default)
  const bool ahead_w = (tn.AHEAD == 1 || env_int("PG_TNP_AHEAD_W", 0) != 0) && W > 1 && LAG + LAGR >= 4 &&
                       ((U == 8 && C == 2 && W == 2) || (U == 16 && C == 1 && (W == 2 || W == 4))) && PF == 2;
  PG_TNP_CASE_A(8, 2, 2, 2, 2, 2); PG_TNP_CASE_A(16, 1, 2, 2, 2, 2); PG_TNP_CASE_A(16, 1, 2, 2, 2, 4);
  PG_TNP_GEOMETRIES;
#undef PG_TNP_CASE
#undef PG_TNP_CASE_D
#undef PG_TNP_CASE_A
#undef PG_TNP_ONE
  pg_set_error("no row-team instantiation for W=%d U=%d C=%d LAG=%d PF=%d LAGR=%d K1=%d PAIR=%d AHEAD=%d%s", W, U, C, LAG, PF, LAGR, k1 ? 1 : 0, pair ? 1 : 0, ahead ? 1 : 0,
               delay ? " with the latency injector" : "");
  return PG_ERR_UNSUPPORTED;
}
#ifdef PG_TN4_FLOAT64_UNIT  // (the instantiations are compiled in two translation units, side by side: pg_gemv_tn4d.hip is this file again)
template pg_status launch_tn_peer<double>(pg_mat*, TNArgs<double>&, int*);
#else
template pg_status launch_tn_peer<float>(pg_mat*, TNArgs<float>&, int*);
#endif

#ifndef PG_TN4_FLOAT64_UNIT
size_t peer_inbox_bytes() { return PEER_RING_BYTES + PEER_SCAL_BYTES; }

// f_out = sum over the devices of *f_local (device order), PG_S_TEAMERR = any device's flag: one launch, no collective
static pg_status peer_exchange(pg_ctx* c, const double* f_local, double* f_out, unsigned aux, double* aux_max_out) {
  pg_row_team& rt = c->rteam;
  PeerScalArgs p;
  p.aux = aux;
  p.aux_max_out = aux_max_out;
  p.spin_limit = rt.tune.SPIN > 0 ? rt.tune.SPIN : TEAM_SPIN_LIMIT;
  p.n = rt.n;
  p.rank = rt.rank;
  rt.scal_epoch = (rt.scal_epoch % 0xFFFFFEu) + 1u;
  p.tag = rt.scal_epoch;
  p.slot = (int)(rt.scal_epoch & 1u);
  for (int q = 0; q < TEAM_MAX; ++q)
    p.inbox[q] = q < rt.n ? (unsigned long long*)((char*)rt.inbox[q] + PEER_RING_BYTES) : nullptr;
  p.f_local = f_local;
  p.f_out = f_out;
  p.team_err = c->dscal + PG_S_TEAMERR;
  hipLaunchKernelGGL(peer_scalars_kernel, dim3(1), dim3(64), 0, c->stream, p);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

pg_status peer_scalar_exchange(pg_ctx* c, const double* f_local, double* f_out) { return peer_exchange(c, f_local, f_out, 0u, nullptr); }

#endif

}  // namespace pgtn

#ifndef PG_TN4_FLOAT64_UNIT
pg_status pg_rteam_sum_scalar(pg_ctx* c, const double* local, double* out) { return pgtn::peer_scalar_exchange(c, local, out); }
#endif
