// Two more geometries of the single-sweep pass (pass TN, see pg_gemv_tn.h), for the column lengths the
// one-workgroup-per-column-group kernel serves badly or not at all:
//
//   gemv_tnw_kernel  SHORT columns (a column fits one wave's registers, <= 8 row groups): every wave works on its own
//                    column groups -- no LDS exchange, no workgroup barrier in the loop.  The C per-column dot products
//                    of a step are reduced ACROSS columns at once (a halving butterfly: v_permlane32_swap /
//                    v_permlane16_swap / DPP row steps; ~C + 6 lane exchanges instead of 6 C), which leaves column c's
//                    total in lane group c, so the forward-backward epilogue (fast_forward_backward.jl:140-142 and :135
//                    of the next iteration) runs ONCE per column, lane-parallel, with coalesced loads / stores of the
//                    n-vectors; v_j returns to the whole wave through v_readlane for the A v accumulation.
//   gemv_tnt_kernel  LONG columns (more row groups than one workgroup's registers hold: m > 32768 in Float32): a TEAM of
//                    workgroups splits the rows of every column.  Each member keeps its slice of r and of the next
//                    residual in registers exactly like the one-workgroup kernel; the members' partial dots of a column
//                    meet through a ring of tagged 8-byte granules in global memory (agent-scope stores / loads, one
//                    granule = {value bits, step tag} so no fence is needed), every member forms the same total in the
//                    same order, applies the prox and accumulates A v on its rows.  With LAG > 0 the exchange is taken
//                    off the critical path: step i posts its partials and consumes the totals of step i - LAG, whose
//                    column tile waited in LDS (128 KiB of the CU's 160 KiB).
//
// Same arithmetic as the reference's statement order up to summation order (benchmark/benchmarks.jl:15-16,
// fast_forward_backward.jl:135-142); both are reached through launch_tn in pg_gemv.hip.
#include <type_traits>

#include <mutex>

#include "pg_gemv_tn.h"

namespace pgtn {
namespace {

// ---------------------------------------------------------------------------------------------------------------
// SHORT columns: one wave per column group
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int U, int C>
struct WTile {
  using V = typename VecOf<T>::type;
  static constexpr int VEC = VecOf<T>::N;
  V col[C][U];
  __device__ __forceinline__ void load(const TNArgs<T>& a, int64_t cg, int lane) {
    const int64_t j0 = cg * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int64_t j = (j0 + c < a.n) ? (j0 + c) : (a.n - 1);
      const T* __restrict__ p = a.A + j * a.ld + lane * VEC;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (u < a.nrg) {
          col[c][u] = nt_load(reinterpret_cast<const V*>(p + (int64_t)u * (WAVE * VEC)));
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) col[c][u][e] = T(0);
        }
      }
    }
  }
};

template <typename T, int U, int C, int WPB, bool DOUBLE_BUFFER>
__global__ __launch_bounds__(WPB * 64) void gemv_tnw_kernel(TNArgs<T> a) {
  using V = typename VecOf<T>::type;
  constexpr int VEC = VecOf<T>::N;
  constexpr int G = 64 / C;  // lanes that hold the same column after the reduction
  static_assert(C >= 2 && C <= 32 && (C & (C - 1)) == 0, "C must be a power of two in 2..32");
  __shared__ V sm_acc[WPB > 1 ? WPB * U * WAVE : 1];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t ncg = (a.n + C - 1) / C;
  const int c_own = lane / G;
  const bool lead = (lane % G) == 0;

  V rk[U], racc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) racc[u][e] = T(0);
    if (u < a.nrg) {
      rk[u] = *reinterpret_cast<const V*>(a.r + (int64_t)u * (WAVE * VEC) + lane * VEC);
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) rk[u][e] = T(0);
    }
  }
  double acc[4] = {0.0, 0.0, 0.0, 0.0};

  auto process = [&](const WTile<T, U, C>& t, int64_t cg) {
    const int64_t j = cg * C + c_own;
    const bool valid = j < a.n;
    const int64_t jc = valid ? j : (a.n - 1);
    const T xj = a.x[jc], zo = a.z_old[jc];  // in flight during the dot products
    T d[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      T s = T(0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) s = fma(t.col[c][u][e], rk[u][e], s);
      }
      d[c] = s;
    }
    cr_stage<T, C, 0>(d, lane);
    T g = d[0];  // = A_j' r for column j = cg * C + lane / G
    if (a.lam_ls != T(1)) g = a.lam_ls * g;
    const T yj = xj - a.gamma * g;  // forward_backward.jl:117 / fast_forward_backward.jl:140
    T zj;                            // :118 / :141
    if (a.g_kind == PG_G_NORML1) {
      T th = a.p0;
      if (a.p0v != nullptr) th = pg_l1w_threshold(a.gamma, a.p0v[jc]);  // per-element weights lam_j
      zj = yj <= -th ? yj + th : (yj >= th ? yj - th : T(0));
    } else if (a.g_kind == PG_G_INDBOX) {
      T lo = a.p0, hi = a.p1;
      if (a.p0v != nullptr) lo = a.p0v[jc], hi = a.p1v[jc];  // per-element bounds
      zj = fmin(hi, fmax(lo, yj));
    } else
      zj = yj;
    const T rj = xj - zj;                                     // :120 / :142
    const T vj = valid ? (a.v_is_res ? rj : zj + a.beta * (zj - zo)) : T(0);      // fast_forward_backward.jl:135 of the next iteration
    if (lead && valid) {
      a.g_out[j] = g;
      a.y[j] = yj;
      a.z_new[j] = zj;
      a.res[j] = rj;
      if (a.v_out != nullptr) a.v_out[j] = vj;
      if (a.g_kind == PG_G_NORML1) acc[0] += a.p0v != nullptr ? (double)a.p0v[j] * fabs((double)zj) : fabs((double)zj);
      acc[1] = pg_maxn(acc[1], fabs((double)rj));
      acc[2] += (double)g * (double)rj;
      acc[3] += (double)rj * (double)rj;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const T vc = pg_readlane(vj, c * G);
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) racc[u][e] = fma(t.col[c][u][e], vc, racc[u][e]);
      }
    }
  };

  // The waves of a workgroup take WPB adjacent column groups per step (one contiguous run of A, WPB * C adjacent outputs);
  // those runs go to the workgroups in chunks of whole output lines (CgMap, pg_gemv_tn.h).  A wave's last step may lie
  // past the end (ncg not a multiple of WPB): its columns are then all invalid -- clamped loads, nothing stored, v = 0.
  const CgMap map((ncg + WPB - 1) / WPB, C * WPB, a.line_cols, blockIdx.x, gridDim.x);
  const int64_t cnt = map.cnt;
  auto at = [&](int64_t i) { return map.at(i) * WPB + wave; };
  if constexpr (DOUBLE_BUFFER) {
    WTile<T, U, C> ta, tb;
    int64_t i = 0;
    if (i < cnt) ta.load(a, at(i), lane);
    while (i < cnt) {
      if (i + 1 < cnt) tb.load(a, at(i + 1), lane);
      process(ta, at(i));
      if (i + 1 >= cnt) break;
      if (i + 2 < cnt) ta.load(a, at(i + 2), lane);
      process(tb, at(i + 1));
      i += 2;
    }
  } else {
    WTile<T, U, C> t;
    for (int64_t i = 0; i < cnt; ++i) {
      t.load(a, at(i), lane);
      process(t, at(i));
    }
  }
  // this workgroup's partial of A v: the waves' accumulators are combined in wave order
  T* part = a.partials + (int64_t)blockIdx.x * a.ld + lane * VEC;
  if constexpr (WPB == 1) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (u < a.nrg) *reinterpret_cast<V*>(part + (int64_t)u * (WAVE * VEC)) = racc[u];
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) sm_acc[(wave * U + u) * WAVE + lane] = racc[u];
    __syncthreads();
    for (int u = wave; u < U; u += WPB) {
      V s = sm_acc[u * WAVE + lane];
#pragma unroll
      for (int w = 1; w < WPB; ++w) {
        const V o = sm_acc[(w * U + u) * WAVE + lane];
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[e] += o[e];
      }
      if (u < a.nrg) *reinterpret_cast<V*>(part + (int64_t)u * (WAVE * VEC)) = s;
    }
  }
  const double ps[4] = {a.gscale, 1.0, 1.0, 1.0};
  grid_reduce_finalize<4, 0x2u, WPB>(acc, a.red_partials, a.red_counter, a.scal_out, ps);
}

template <typename T, int U, int C, int WPB, bool DB>
pg_status launch_tnw(pg_mat* A, TNArgs<T>& a, int* blocks_out) {
  pg_ctx* c = A->ctx;
  const int64_t ncg = (A->n + C - 1) / C;
  // waves per CU: every wave keeps one (two when double-buffered) C * U KiB tile in flight
  const int wpc = env_int("PG_TNW_WAVES_PER_CU", DB ? 16 : 4);
  int64_t blocks = ((int64_t)c->num_cu * wpc + WPB - 1) / WPB;
  if (env_int("PG_TN_BLOCKS", 0) > 0) blocks = env_int("PG_TN_BLOCKS", 0);
  if (blocks > (ncg + WPB - 1) / WPB) blocks = (ncg + WPB - 1) / WPB;
  if (blocks > PG_RED_MAX_BLOCKS) blocks = PG_RED_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  PG_TRY(ensure_partials(A, (int)blocks));
  a.partials = (T*)A->partials;
  *blocks_out = (int)blocks;
  pg_prof_scope prof(c, PG_K_GEMV_TN);
  hipLaunchKernelGGL((gemv_tnw_kernel<T, U, C, WPB, DB>), dim3((unsigned)blocks), dim3(WPB * 64), 0, c->stream, a);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// MEDIUM columns (9..32 row groups): the waves of a workgroup SHARE every column group like gemv_tn_kernel, with the
// column-parallel reduction and the lane-parallel epilogue of gemv_tnw_kernel: per step each wave reduces its C partial
// dots across columns (C + 6 lane exchanges), the lane groups' leaders leave them in LDS, and after the one workgroup
// barrier of the step every lane sums the WAVES partials of ITS column, applies the prox once and hands v_j back through
// v_readlane.  (gemv_tn_kernel: 6 C shuffles per wave and the epilogue of all C columns in every lane.)
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int U, int C, int WAVES, bool DOUBLE_BUFFER>
__global__ __launch_bounds__(WAVES * 64) void gemv_tnc_kernel(TNArgs<T> a) {
  using V = typename VecOf<T>::type;
  constexpr int VEC = VecOf<T>::N;
  constexpr int G = 64 / C;
  static_assert(C >= 2 && C <= 32 && (C & (C - 1)) == 0, "C must be a power of two in 2..32");
  __shared__ T sm_part[2][WAVES][C];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t ncg = (a.n + C - 1) / C;
  const int c_own = lane / G;
  const bool lead = (lane % G) == 0;

  V rk[U], racc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int rg = TN_RG(u, wave, U, WAVES);
#pragma unroll
    for (int e = 0; e < VEC; ++e) racc[u][e] = T(0);
    if (rg < a.nrg) {
      rk[u] = *reinterpret_cast<const V*>(a.r + (int64_t)rg * (WAVE * VEC) + lane * VEC);
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) rk[u][e] = T(0);
    }
  }
  double acc[4] = {0.0, 0.0, 0.0, 0.0};

  auto process = [&](const TNTile<T, U, C, WAVES>& t, int64_t cg, int buf) {
    const int64_t j = cg * C + c_own;
    const bool valid = j < a.n;
    const int64_t jc = valid ? j : (a.n - 1);
    const T xj = a.x[jc], zo = a.z_old[jc];  // in flight during the dot products
    T d[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      T s = T(0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) s = fma(t.col[c][u][e], rk[u][e], s);
      }
      d[c] = s;
    }
    cr_stage<T, C, 0>(d, lane);
    if (lead) sm_part[buf][wave][c_own] = d[0];
    __syncthreads();
    T g = sm_part[buf][0][c_own];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) g += sm_part[buf][w][c_own];  // wave order: every wave forms the same bits
    if (a.lam_ls != T(1)) g = a.lam_ls * g;
    const T yj = xj - a.gamma * g;  // forward_backward.jl:117 / fast_forward_backward.jl:140
    T zj;                            // :118 / :141
    if (a.g_kind == PG_G_NORML1) {
      T th = a.p0;
      if (a.p0v != nullptr) th = pg_l1w_threshold(a.gamma, a.p0v[jc]);  // per-element weights lam_j
      zj = yj <= -th ? yj + th : (yj >= th ? yj - th : T(0));
    } else if (a.g_kind == PG_G_INDBOX) {
      T lo = a.p0, hi = a.p1;
      if (a.p0v != nullptr) lo = a.p0v[jc], hi = a.p1v[jc];  // per-element bounds
      zj = fmin(hi, fmax(lo, yj));
    } else
      zj = yj;
    const T rj = xj - zj;                                 // :120 / :142
    const T vj = valid ? (a.v_is_res ? rj : zj + a.beta * (zj - zo)) : T(0);  // fast_forward_backward.jl:135 of the next iteration
    if (wave == 0 && lead && valid) {
      a.g_out[j] = g;
      a.y[j] = yj;
      a.z_new[j] = zj;
      a.res[j] = rj;
      if (a.v_out != nullptr) a.v_out[j] = vj;
      if (a.g_kind == PG_G_NORML1) acc[0] += a.p0v != nullptr ? (double)a.p0v[j] * fabs((double)zj) : fabs((double)zj);
      acc[1] = pg_maxn(acc[1], fabs((double)rj));
      acc[2] += (double)g * (double)rj;
      acc[3] += (double)rj * (double)rj;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const T vc = pg_readlane(vj, c * G);
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) racc[u][e] = fma(t.col[c][u][e], vc, racc[u][e]);
      }
    }
  };

  const CgMap map(ncg, C, a.line_cols, blockIdx.x, gridDim.x);
  const int64_t cnt = map.cnt;
  auto at = [&](int64_t i) { return map.at(i); };
  if constexpr (DOUBLE_BUFFER) {
    TNTile<T, U, C, WAVES> ta, tb;
    int64_t i = 0;
    if (i < cnt) ta.load(a, at(i), wave, lane);
    while (i < cnt) {
      if (i + 1 < cnt) tb.load(a, at(i + 1), wave, lane);
      process(ta, at(i), 0);
      if (i + 1 >= cnt) break;
      if (i + 2 < cnt) ta.load(a, at(i + 2), wave, lane);
      process(tb, at(i + 1), 1);
      i += 2;
    }
  } else {
    TNTile<T, U, C, WAVES> t;
    int buf = 0;
    for (int64_t i = 0; i < cnt; ++i) {
      const int64_t cg = at(i);
      t.load(a, cg, wave, lane);
      process(t, cg, buf);
      buf ^= 1;
    }
  }
  T* part = a.partials + (int64_t)blockIdx.x * a.ld + lane * VEC;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int rg = TN_RG(u, wave, U, WAVES);
    if (rg < a.nrg) *reinterpret_cast<V*>(part + (int64_t)rg * (WAVE * VEC)) = racc[u];
  }
  const double ps[4] = {a.gscale, 1.0, 1.0, 1.0};
  grid_reduce_finalize<4, 0x2u, WAVES>(acc, a.red_partials, a.red_counter, a.scal_out, ps);
}

template <typename T, int U, int C, int WAVES, bool DB>
pg_status launch_tnc(pg_mat* A, TNArgs<T>& a, int* blocks_out) {
  pg_ctx* c = A->ctx;
  const int64_t ncg = (A->n + C - 1) / C;
  int64_t blocks = (int64_t)c->num_cu * env_int("PG_TNC_BLOCKS_PER_CU", DB ? (WAVES >= 8 ? 1 : 2) : (WAVES >= 8 ? 1 : (WAVES == 4 ? 1 : 2)));
  if (env_int("PG_TN_BLOCKS", 0) > 0) blocks = env_int("PG_TN_BLOCKS", 0);
  if (blocks > ncg) blocks = ncg;
  if (blocks > PG_RED_MAX_BLOCKS) blocks = PG_RED_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  PG_TRY(ensure_partials(A, (int)blocks));
  a.partials = (T*)A->partials;
  *blocks_out = (int)blocks;
  pg_prof_scope prof(c, PG_K_GEMV_TN);
  hipLaunchKernelGGL((gemv_tnc_kernel<T, U, C, WAVES, DB>), dim3((unsigned)blocks), dim3(WAVES * 64), 0, c->stream, a);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

#include "pg_gemv_tnt.h"  // gemv_tnt_kernel: the team sweep (shared with pg_gemv_tn4.hip)

template <typename T, int U, int C, int LAG, int PF, int WAVES, int LAGR = 0>
pg_status launch_tnt(pg_mat* A, TNArgs<T>& a, int* blocks_out) {
  pg_ctx* c = A->ctx;
  constexpr int G = (int)sizeof(T) / 4;
  const int64_t ncg = (A->n + C - 1) / C;
  // Four-wave members hold up to 64 row groups; the rows are dealt evenly over the team's waves (ueff each) and the kernel
  // is instantiated for U = ueff exactly (9..16), so that a column length that only partly fills the last member costs
  // neither idle workgroups nor repeated loads: 50000 rows are 4 members x 4 waves x 13 row groups, not 3 full members + 4
  // row groups.  (Eight-wave geometries, kept for the tuning sweeps: U row groups per wave, filled in order.)
  const int per_member = WAVES == 4 ? TEAM_MEMBER_RG : U * WAVES;
  int TM = (a.nrg + per_member - 1) / per_member;
  if (env_int("PG_TN_TEAM", 0) > TM) TM = env_int("PG_TN_TEAM", 0);  // experiments: more (partly idle) members
  if (TM < 1) TM = 1;
  if (TM > TEAM_MAX) {
    pg_set_error("the single-sweep pass covers columns of at most %d rows", (int)(TEAM_MAX * per_member * (1024 / sizeof(T))));
    return PG_ERR_UNSUPPORTED;
  }
  // every member must be resident at once (they wait for each other): one workgroup per CU at most
  int64_t nteams = c->num_cu / TM;
  if (env_int("PG_TN_TEAMS", 0) > 0 && env_int("PG_TN_TEAMS", 0) < nteams) nteams = env_int("PG_TN_TEAMS", 0);
  if (nteams > ncg) nteams = ncg;
  if (nteams < 1) nteams = 1;
  if (nteams * TM > c->num_cu) {
    pg_set_error("a team of %d workgroups does not fit the device's %d compute units", TM, c->num_cu);
    return PG_ERR_UNSUPPORTED;
  }
  const size_t xch_bytes = (size_t)nteams * TEAM_RING * (size_t)(TEAM_MAX * C * G) * sizeof(unsigned long long);
  bool fresh_ring = false;
  if (A->xch == nullptr || A->xch_bytes < xch_bytes) {
    fresh_ring = true;
    if (A->xch) {
      PG_HIP(hipStreamSynchronize(c->stream));
      PG_HIP(hipFree(A->xch));
      A->xch = nullptr;
    }
    hipError_t e = hipMalloc(&A->xch, xch_bytes);
    if (e != hipSuccess) {
      pg_set_error("hipMalloc(%zu) for the team exchange ring failed: %s", xch_bytes, hipGetErrorString(e));
      return PG_ERR_ALLOC;
    }
    A->xch_bytes = xch_bytes;
  }
  const long long layout = ((long long)nteams << 20) | ((long long)TM << 8) | (long long)(C * G);  // (the lag is not part of it: a slot's tag says which step it holds)
  if (layout != A->xch_layout) fresh_ring = true;
  A->xch_layout = layout;
  if (fresh_ring) PG_HIP(hipMemsetAsync(A->xch, 0, xch_bytes, c->stream));
  PG_TRY(ensure_partials(A, (int)nteams));
  a.partials = (T*)A->partials;
  a.team_size = TM;
  a.ueff = WAVES == 4 ? (a.nrg + TM * WAVES - 1) / (TM * WAVES) : U;
  a.deal_even = WAVES == 4 && env_int("PG_TNT_EVEN", 1) ? 1 : 0;
  if (a.ueff > U) {
    pg_set_error("gemv_tnt<U = %d> launched for %d row groups per wave", U, a.ueff);
    return PG_ERR_INVALID;
  }
  const int64_t steps = (ncg + nteams - 1) / nteams;
  if (steps + 1 >= (1 << 24)) {
    pg_set_error("the long-column sweep covers at most 2^24 column groups per team (%lld here)", (long long)steps);
    return PG_ERR_UNSUPPORTED;
  }
  a.nteams = (int)nteams;
  a.xch = (unsigned long long*)A->xch;
  a.team_err = c->dscal + PG_S_TEAMERR;
  if (c->capturing) {
    // a recorded launch replays with the kernel arguments baked in, so its tags cannot carry a fresh epoch: every replay
    // would meet its own granules of the previous replay (same step, same tag) in the ring.  The recorded body therefore
    // zeroes the ring itself before the sweep (a memset node, as before round 3) under the reserved epoch 255, which no
    // uncaptured launch uses.
    PG_HIP(hipMemsetAsync(A->xch, 0, xch_bytes, c->stream));
    a.tag_base = 255u << 24;
  } else {
    A->xch_epoch = (A->xch_epoch % 254u) + 1u;  // 1 .. 254: never the all-zero tag of a fresh ring, never a replay's 255
    a.tag_base = A->xch_epoch << 24;
  }
#ifdef PG_TNT_EXPERIMENT
  a.dbg = env_int("PG_TNT_DBG", 0);
#endif
  *blocks_out = (int)nteams;
  // LDS for the parked tiles: LAG slots of WAVES * C * U KiB on top of the kernel's static LDS (the dot exchange and the grid
  // reduction's scratch); anything beyond the default 64 KiB limit is opted into once per device
  const size_t lds = (size_t)LAG * WAVES * C * U * 1024;
  const void* kern = reinterpret_cast<const void*>(&gemv_tnt_kernel<T, U, C, WAVES, LAG, PF, false, LAGR>);
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
  // Every member of every team must be resident at once (they wait for each other).  A cooperative launch makes that the
  // runtime's promise instead of this file's assumption: it is refused when the grid cannot be co-resident with this
  // kernel's registers and LDS, and the cooperative queue keeps a second cooperative grid from interleaving with it.
  // What it cannot promise is that nothing ELSE holds compute units for the whole duration (another process, a kernel of
  // another stream that itself waits for this one): the members' waits are bounded for that case (TEAM_SPIN_LIMIT ->
  // PG_S_TEAMERR -> PG_ERR_TIMEOUT at the next scalar read-back, which the iterations turn into a two-sweep retry).
  unsigned grid = (unsigned)(nteams * TM);
  c->team_launches++;
  if (c->test_team_fault > 0 && c->team_launches == c->test_team_fault && TM > 1) {
    if (c->test_team_fault_kind == 1) {  // test hook: this launch is refused, as a cooperative launch that does not fit would be
      pg_set_error("cooperative launch of the long-column sweep was refused (injected by pg_ctx_test_team_fault)");
      return PG_ERR_UNSUPPORTED;
    }
    grid -= 1;  // test hook: the last member of the last team is never started, its team-mates time out
  }
  pg_prof_scope prof(c, PG_K_GEMV_TN);
  if (c->team_plain_launch || c->capturing) {  // (stream capture records plain launches only)
    hipLaunchKernelGGL((gemv_tnt_kernel<T, U, C, WAVES, LAG, PF, false, LAGR>), dim3(grid), dim3(WAVES * 64), lds, c->stream, a);
    PG_LAUNCH_CHECK();
    return PG_OK;
  }
  // the probe of the previous cooperative sweep, if it has finished: was it at streaming rate?
  constexpr double COOP_SLOW_BYTES_PER_S = 4.0e12;  // the sweep streams 6.9-7.2 TB/s; next to a foreign cooperative queue 3.1
  if (c->coop_probe_bytes > 0 && hipEventQuery(c->coop_probe[1]) == hipSuccess) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->coop_probe[0], c->coop_probe[1]) == hipSuccess && ms > 0.f) {
      const double rate = c->coop_probe_bytes / (1e-3 * (double)ms);
      c->coop_slow_in_a_row = rate < COOP_SLOW_BYTES_PER_S ? c->coop_slow_in_a_row + 1 : 0;
      if (c->coop_slow_in_a_row == 2) {
        // Measured (profiles/r4_coop_probe.md): switching THIS context to plain launches now does not bring the rate back -- a
        // process that has used its cooperative queue stays in the device's alternation -- so the library only says what it
        // sees; the remedy is to start the process with PG_TN_TEAM_PLAIN=1 (read at context creation).
        c->coop_slow = true;
        fprintf(stderr, "libproxgrad_hip: the cooperative team sweep ran at %.1f TB/s twice in a row (it streams ~7): another process "
                        "holds a cooperative queue on this device and the device alternates between the two.  Results are unaffected; "
                        "start this process with PG_TN_TEAM_PLAIN=1 to launch the sweep plainly (full rate next to an idle process)\n",
                rate / 1e12);
      }
    }
    c->coop_probe_bytes = 0;
  } else {
    (void)hipGetLastError();  // hipErrorNotReady of the query is not an error
  }
  const double sweep_bytes = (double)A->m * (double)A->n * sizeof(T);
  const bool probe = c->coop_probes_left > 0 && c->coop_probe_bytes == 0 && sweep_bytes >= (double)(1u << 30) && !c->profiling;
  if (probe) {
    if (c->coop_probe[0] == nullptr) {
      PG_HIP(hipEventCreate(&c->coop_probe[0]));
      PG_HIP(hipEventCreate(&c->coop_probe[1]));
    }
    PG_HIP(hipEventRecord(c->coop_probe[0], c->stream));
  }
  void* args[1] = {(void*)&a};
  hipError_t e;
  {
    std::lock_guard<std::mutex> lock(pg_coop_launch_mutex());  // (see pg_internal.h: concurrent cooperative launches crash the process at exit)
    e = hipLaunchCooperativeKernel(kern, dim3(grid), dim3(WAVES * 64), args, (unsigned)lds, c->stream);
  }
  if (probe && e == hipSuccess) {
    PG_HIP(hipEventRecord(c->coop_probe[1], c->stream));
    c->coop_probe_bytes = sweep_bytes;
    c->coop_probes_left--;
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    pg_set_error("cooperative launch of the long-column sweep (%u workgroups of %d threads, %zu bytes of LDS) was refused: %s",
                 grid, WAVES * 64, lds, hipGetErrorString(e));
    return e == hipErrorCooperativeLaunchTooLarge || e == hipErrorInvalidConfiguration || e == hipErrorInvalidValue ||
                   e == hipErrorNotSupported
               ? PG_ERR_UNSUPPORTED
               : PG_ERR_HIP;
  }
  return PG_OK;
}

}  // namespace

bool tn_wave_covers(int nrg) { return nrg >= 1 && nrg <= 8; }
bool tn_team_covers(int nrg) { return nrg >= 1 && nrg <= TEAM_MAX * TEAM_MEMBER_RG; }

// Tunables (environment, for experiments): PG_TNW_C, PG_TNW_WPB (waves per workgroup), PG_TNW_DB (0 / 1),
// PG_TNW_WAVES_PER_CU.
template <typename T>
pg_status launch_tn_wave(pg_mat* A, TNArgs<T>& a, int* blocks_out) {
  const int nrg = a.nrg;
  int U = 1;
  while (U < nrg) U *= 2;
  // defaults from the sweeps in profiles/r2_tune_tn_short_columns.log: 32 KiB tiles (C * U = 32), four-wave workgroups;
  // columns of >= 3 row groups: ONE tile per wave and one workgroup per CU (6.7 / 6.9 TB/s at 1024 / 2048 rows, Float32);
  // shorter ones: two tiles per wave, four workgroups per CU (6.3 TB/s at 256 and 512 rows)
  int C = env_int("PG_TNW_C", 32 / U);
  const int WPB = env_int("PG_TNW_WPB", 4);
  const int DB = env_int("PG_TNW_DB", U >= 4 ? 0 : 1);
  if (sizeof(T) == 8 && C > 16) C = 16;
#define PG_TNW_CASE(UU, CC, WW, DD) \
  if (U == UU && C == CC && WPB == WW && DB == DD) return launch_tnw<T, UU, CC, WW, (DD != 0)>(A, a, blocks_out)
#define PG_TNW_UC(UU, CC)   \
  PG_TNW_CASE(UU, CC, 1, 1); \
  PG_TNW_CASE(UU, CC, 1, 0); \
  PG_TNW_CASE(UU, CC, 2, 0); \
  PG_TNW_CASE(UU, CC, 8, 0); \
  PG_TNW_CASE(UU, CC, 4, 1); \
  PG_TNW_CASE(UU, CC, 4, 0)
  PG_TNW_UC(1, 8);
  PG_TNW_UC(1, 16);
  PG_TNW_UC(1, 32);
  PG_TNW_UC(2, 4);
  PG_TNW_UC(2, 8);
  PG_TNW_UC(2, 16);
  PG_TNW_UC(4, 2);
  PG_TNW_UC(4, 4);
  PG_TNW_UC(4, 8);
  PG_TNW_UC(8, 2);
  PG_TNW_UC(8, 4);
#undef PG_TNW_UC
#undef PG_TNW_CASE
  pg_set_error("no gemv_tnw instantiation for U=%d C=%d WPB=%d DB=%d", U, C, WPB, DB);
  return PG_ERR_UNSUPPORTED;
}

bool tn_coop_covers(int nrg) { return nrg >= 3 && nrg <= 32; }

// Tunables (environment, for experiments): PG_TNC_WAVES, PG_TNC_C, PG_TNC_DB, PG_TNC_BLOCKS_PER_CU.
template <typename T>
pg_status launch_tn_coop(pg_mat* A, TNArgs<T>& a, int* blocks_out) {
  const int nrg = a.nrg;
  // measured (profiles/r2_tune_tn_mid_columns.log): eight waves of U = 2 with C = 16 columns per step, one tile, one workgroup
  // per CU -- 4096 x 2^19: 6.68 TB/s against 6.37 for gemv_tn's four waves of U = 4; 3072 rows: 6.62 against 6.03;
  // 17..24 row groups: eight waves of U = 4, C = 8: 7.03 / 6.83 / 6.96 TB/s against gemv_tn's 6.27 / 6.53 / 6.88; from 28
  // row groups on the two kernels are level (8192 x 2^18: 6.85 / 6.88) and gemv_tn stays
  const int W = env_int("PG_TNC_WAVES", 8);
  int U = 1;
  while (U * W < nrg) U *= 2;
  // Chunks of whole output lines (CgMap) pay here as well -- 3072 / 5120 / 6144 rows: 0.836-0.857 of 8 TB/s against 0.80-0.84
  // dealt one by one -- with ONE measured exception: columns of exactly 16 row groups (4096 rows f32), 0.79-0.80 against
  // 0.84 (profiles/r3_tune_tn_mid_columns.log, two rounds, two boxes); that length keeps the round-robin deal.
  if (nrg == 16 && env_int("PG_TN_LINE_COLS", 0) == 0) a.line_cols = 1;
  const int C = env_int("PG_TNC_C", 32 / U);
  const int DB = env_int("PG_TNC_DB", 0);
#define PG_TNC_CASE(UU, CC, WW, DD) \
  if (U == UU && C == CC && W == WW && DB == DD) return launch_tnc<T, UU, CC, WW, (DD != 0)>(A, a, blocks_out)
#define PG_TNC_UCW(UU, CC, WW) \
  PG_TNC_CASE(UU, CC, WW, 0); \
  PG_TNC_CASE(UU, CC, WW, 1)
  PG_TNC_UCW(4, 8, 4);
  PG_TNC_UCW(4, 4, 4);
  PG_TNC_UCW(8, 4, 2);
  PG_TNC_UCW(8, 2, 2);
  PG_TNC_UCW(2, 16, 8);
  PG_TNC_UCW(2, 8, 8);
  PG_TNC_UCW(4, 8, 8);
  PG_TNC_UCW(4, 4, 8);
  PG_TNC_UCW(8, 4, 4);
  PG_TNC_UCW(8, 2, 4);
  PG_TNC_UCW(2, 16, 4);
  PG_TNC_UCW(2, 8, 4);
#undef PG_TNC_UCW
#undef PG_TNC_CASE
  pg_set_error("no gemv_tnc instantiation for U=%d C=%d WAVES=%d DB=%d", U, C, W, DB);
  return PG_ERR_UNSUPPORTED;
}

// Tunables (environment, for experiments): PG_TNT_WAVES, PG_TNT_U, PG_TNT_C, PG_TNT_LAG, PG_TNT_PF, PG_TN_TEAM (members per
// team), PG_TN_TEAMS.
template <typename T>
pg_status launch_tn_team(pg_mat* A, TNArgs<T>& a, int* blocks_out) {
  // The exchange costs about two step times (the post and the poll each queue behind a step's worth of loads in their
  // CU's memory pipeline) plus the fabric: the totals of step i are consumed LAG steps later.  LAG * C * U * WAVES <= 128:
  // the parked tiles fill 128 KiB of LDS.  A member is FOUR waves of U = 16 row groups (one wave per SIMD, 16 KiB of a
  // column per wave and step, the full 512 registers): 131072 rows 7.01-7.05 TB/s, 65536 rows 7.03-7.06; eight waves of
  // U = 8 (the same rows per member, half the bytes per wave and step): 6.73-6.75 / 6.41-6.56; eight waves of U = 4 with
  // two columns per step: 6.20 / 6.79-6.81; LAG = 1: 5.05-5.61, LAG = 0: 4.62-5.49 TB/s (profiles/r2_tune_tn_team.log).
  const int W = env_int("PG_TNT_WAVES", 4);
  int U = env_int("PG_TNT_U", W == 4 ? 0 : 8);
  if (U == 0) {  // four-wave members: the smallest instantiation that holds the even deal
    // The fewest members that hold the column.  (Measured, profiles/r3_team_pattern.md: more members of fewer row groups each,
    // where that leaves less padding -- 50000 rows as 5 x 4 x 10 instead of 4 x 4 x 13, 150000 rows as 15 x 4 x 10 instead of
    // 10 x 4 x 15 -- is slower every time, 0.816 against 0.829 and 0.735 against 0.842 of 8 TB/s: shorter steps, same exchange.)
    int TM = (a.nrg + TEAM_MEMBER_RG - 1) / TEAM_MEMBER_RG;
    if (env_int("PG_TN_TEAM", 0) > TM) TM = env_int("PG_TN_TEAM", 0);
    U = (a.nrg + TM * 4 - 1) / (TM * 4);
    if (U < 8) U = 8;
    if (U > 16) U = 16;  // (more than 16 members' worth of rows: launch_tnt reports it)
  }
  const int C = env_int("PG_TNT_C", U == 4 ? 2 : 1), LAG = env_int("PG_TNT_LAG", 2);
  // tiles in flight ahead of the one being consumed
  const int PF = env_int("PG_TNT_PF", U == 4 ? 1 : 2);
  const int LAGR = env_int("PG_TNT_LAGR", 0);  // lag tiles in registers (pg_gemv_tnt.h; measured +0.2 % at 131072 rows: not the default here)
#define PG_TNT_CASE_X(UU, CC, LL, PP, WW, RR) \
  if (U == UU && C == CC && LAG == LL && PF == PP && W == WW && LAGR == RR) return launch_tnt<T, UU, CC, LL, PP, WW, RR>(A, a, blocks_out)
#define PG_TNT_CASE(UU, CC, LL, PP, WW) PG_TNT_CASE_X(UU, CC, LL, PP, WW, 0)
  if constexpr (sizeof(T) == 4) {
    PG_TNT_CASE_X(16, 1, 2, 2, 4, 1);
    PG_TNT_CASE_X(16, 1, 0, 2, 4, 2);  // no LDS round trip at all: the two waiting tiles stay in registers
  }
  PG_TNT_CASE(16, 1, 2, 2, 4);
  PG_TNT_CASE(15, 1, 2, 2, 4);
  PG_TNT_CASE(14, 1, 2, 2, 4);
  PG_TNT_CASE(13, 1, 2, 2, 4);
  PG_TNT_CASE(12, 1, 2, 2, 4);
  PG_TNT_CASE(11, 1, 2, 2, 4);
  PG_TNT_CASE(10, 1, 2, 2, 4);
  PG_TNT_CASE(9, 1, 2, 2, 4);
  PG_TNT_CASE(8, 1, 2, 2, 4);
  PG_TNT_CASE(16, 1, 2, 1, 4);
  PG_TNT_CASE(16, 1, 1, 2, 4);
  PG_TNT_CASE(16, 1, 0, 2, 4);
  PG_TNT_CASE(8, 1, 2, 2, 8);
  PG_TNT_CASE(8, 1, 2, 1, 8);
  PG_TNT_CASE(4, 2, 2, 1, 8);
#undef PG_TNT_CASE
#undef PG_TNT_CASE_X
  pg_set_error("no gemv_tnt instantiation for WAVES=%d U=%d C=%d LAG=%d PF=%d LAGR=%d", W, U, C, LAG, PF, LAGR);
  return PG_ERR_UNSUPPORTED;
}

template pg_status launch_tn_wave<float>(pg_mat*, TNArgs<float>&, int*);
template pg_status launch_tn_wave<double>(pg_mat*, TNArgs<double>&, int*);
template pg_status launch_tn_coop<float>(pg_mat*, TNArgs<float>&, int*);
template pg_status launch_tn_coop<double>(pg_mat*, TNArgs<double>&, int*);
template pg_status launch_tn_team<float>(pg_mat*, TNArgs<float>&, int*);
template pg_status launch_tn_team<double>(pg_mat*, TNArgs<double>&, int*);

}  // namespace pgtn
