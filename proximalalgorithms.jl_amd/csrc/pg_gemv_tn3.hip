// Columns of 29 .. 128 row groups (7424 .. 32768 rows in Float32, the headline's 16384 among them): gemv_tnm_kernel, the
// one-workgroup sweep with a row-group count per wave U that fits the column exactly (9 .. 16 rather than the next power
// of two), branch-free tile loads and x_j / z_old_j issued ahead of the tile, compiled in its own translation unit.
// Measurements behind it: scripts/tile_pattern.hip and scripts/r3_mid_sweep.py (profiles/r3_mid_columns_counters.md,
// profiles/r3_tune_tn_mid_columns.log).
#include <mutex>

#include "pg_gemv_tn.h"

namespace pgtn {

namespace {
__device__ __forceinline__ int wave_of(unsigned tid) { return (int)(tid >> 6); }

// ---------------------------------------------------------------------------------------------------------------
// gemv_tnm_kernel: the one-workgroup sweep re-cut for these lengths.  Same work split, same summation order and the same
// epilogue as gemv_tn_kernel (pg_gemv_tn.h) -- wave w owns row groups w * U .. w * U + U - 1, the C column dots of a step
// meet in LDS behind one barrier -- with the two things the disassembly of gemv_tn_kernel shows in the way at 8-32 KiB
// columns taken out (profiles/r3_mid_columns_counters.md):
//   * every tile load sat behind its own scalar compare-and-branch (row group < nrg) with the address arithmetic in
//     between, ~10 instructions per 1 KiB load.  Here row groups past the end are CLAMPED to the last one (its lines are
//     in the cache; the matching r entries are zero, so they add nothing) and the C * U loads of a tile go out back to back;
//   * x_j and z_old_j were loaded AFTER the tile had arrived (s_waitcnt vmcnt(0) in front of them), so every step paid
//     one more memory round trip between its barrier and its epilogue with nothing in flight.  Here they are issued
//     BEFORE the tile's loads (vector loads return in order: they are back first), between scheduling fences.  (Scalar
//     loads were tried: through the constant address space the compiler sinks them to the instruction in front of the
//     barrier, fences or not; as hand-written s_load asm the destination register is reused before the data lands.)
// NT = 1: one register tile (load, wait, consume); NT = 2: the next tile's loads are in flight while this one is consumed.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int U, int C, int WAVES, int NT>
__global__ __launch_bounds__(WAVES * 64) void gemv_tnm_kernel(TNArgs<T> a) {
  using V = typename VecOf<T>::type;
  constexpr int VEC = VecOf<T>::N;
  __shared__ T sm_dot[2][C][WAVES];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t ncg = (a.n + C - 1) / C;
  V rk[U], racc[U];
  int rgo[U];  // element offset of this wave's row group u within a column (clamped)
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int rg = wave * U + u;
    rgo[u] = (rg < a.nrg ? rg : a.nrg - 1) * (WAVE * VEC);
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

  struct Tile {
    V col[C][U];
    T xs[C], zos[C];
  };
  auto load = [&](Tile& t, int64_t cg) __attribute__((always_inline)) {
    const int64_t j0 = cg * C;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int64_t jc = (j0 + c < a.n) ? (j0 + c) : (a.n - 1);
      t.xs[c] = a.x[jc];
      t.zos[c] = a.z_old[jc];
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int64_t jc = (j0 + c < a.n) ? (j0 + c) : (a.n - 1);
      const T* __restrict__ p = a.A + jc * a.ld;  // wave-uniform base (scalar registers) + one shared per-lane offset
#pragma unroll
      for (int u = 0; u < U; ++u) t.col[c][u] = nt_load(reinterpret_cast<const V*>(p + rgo[u]) + lane);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto process = [&](const Tile& t, int64_t cg, int buf) __attribute__((always_inline)) {
    const int64_t j0 = cg * C;
    T dot[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      T d = T(0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) d = fma(t.col[c][u][e], rk[u][e], d);
      }
      dot[c] = wave_allsum(d);
    }
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < C; ++c) sm_dot[buf][c][wave] = dot[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < C; ++c) {
      T g = sm_dot[buf][c][0];
#pragma unroll
      for (int w = 1; w < WAVES; ++w) g += sm_dot[buf][c][w];
      const int64_t j = j0 + c;
      const bool valid = j < a.n;
      if (a.lam_ls != T(1)) g = a.lam_ls * g;
      const T xj = t.xs[c], zo = t.zos[c];
      const T yj = xj - a.gamma * g;  // forward_backward.jl:117 / fast_forward_backward.jl:140
      T zj;                            // :118 / :141
      if (a.g_kind == PG_G_NORML1) {
        T th = a.p0;
        if (a.p0v != nullptr) th = pg_l1w_threshold(a.gamma, a.p0v[valid ? j : a.n - 1]);  // per-element weights lam_j
        zj = yj <= -th ? yj + th : (yj >= th ? yj - th : T(0));
      } else if (a.g_kind == PG_G_INDBOX) {
        T lo = a.p0, hi = a.p1;
        if (a.p0v != nullptr) lo = a.p0v[valid ? j : a.n - 1], hi = a.p1v[valid ? j : a.n - 1];  // per-element bounds
        zj = fmin(hi, fmax(lo, yj));
      } else
        zj = yj;
      const T rj = xj - zj;                                   // :120 / :142
      const T vj = valid ? (a.v_is_res ? rj : zj + a.beta * (zj - zo)) : T(0);    // fast_forward_backward.jl:135 of the next iteration
      if ((int)threadIdx.x == c && valid) {
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
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) racc[u][e] = fma(t.col[c][u][e], vj, racc[u][e]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) asm volatile("" : "+v"(racc[u]));  // see gemv_tn_kernel: keeps this tile from living on
  };

  const CgMap map(ncg, C, a.line_cols, blockIdx.x, gridDim.x);
  const int64_t cnt = map.cnt;
  auto at = [&](int64_t i) { return map.at(i); };
  if constexpr (NT == 2) {
    Tile ta, tb;
    int64_t i = 0;
    if (i < cnt) load(ta, at(i));
    while (i < cnt) {
      if (i + 1 < cnt) load(tb, at(i + 1));
      process(ta, at(i), 0);
      if (i + 1 >= cnt) break;
      if (i + 2 < cnt) load(ta, at(i + 2));
      process(tb, at(i + 1), 1);
      i += 2;
    }
  } else {
    Tile t;
    int buf = 0;
    for (int64_t i = 0; i < cnt; ++i) {
      load(t, at(i));
      process(t, at(i), buf);
      buf ^= 1;
    }
  }
  T* part = a.partials + (int64_t)blockIdx.x * a.ld + lane * VEC;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int rg = wave * U + u;
    if (rg < a.nrg) *reinterpret_cast<V*>(part + (int64_t)rg * (WAVE * VEC)) = racc[u];
  }
  const double ps[4] = {a.gscale, 1.0, 1.0, 1.0};
  grid_reduce_finalize<4, 0x2u, WAVES>(acc, a.red_partials, a.red_counter, a.scal_out, ps);
}

// ---------------------------------------------------------------------------------------------------------------
// gemv_tnm_pair_kernel: TWO instances of the sweep on ONE read of A -- the same matrix, step size and g, two pairs (r, x).
// ZeroFPR's line search (zerofpr.jl:200-217) evaluates x = xbar_prev + tau d for tau = 1, 1/2, ... and every trial point is a
// sweep of its own (A' grad f(A x), the forward-backward step, and A xbar for the next iteration): with the trial points of tau
// and tau / 2 carried through the same register tile a rejected first trial costs no second read.  Per
// column and instance the arithmetic is gemv_tnm_kernel's, statement for statement (per-wave fma chain, fixed-order wave sum, the
// waves' partials in wave order); with eight waves per column where the single sweep has four the partial sums group differently,
// so an instance's results equal a single sweep's to the last bits of the working precision, not bit for bit.
// Second instance: TNPair (its r, x and outputs); the first one's are TNArgs' own.  Scalars: eight slots from a.scal_out.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
struct TNPair {
  const T* r;  // [ld]
  const T* x;  // [n]
  T *g_out, *y, *z_new, *res;  // [n] each
  T* partials;                 // [gridDim.x][ld]
};

template <typename T, int U, int C, int WAVES, int NT>
__global__ __launch_bounds__(WAVES * 64) void gemv_tnm_pair_kernel(TNArgs<T> a, TNPair<T> b) {
  using V = typename VecOf<T>::type;
  constexpr int VEC = VecOf<T>::N;
  __shared__ T sm_dot[2][2][C][WAVES];
  // Both instances' slices of r live in LDS (2 U KiB per wave, each wave reads only its own: no barrier): with the slices and both
  // accumulator sets in registers beside two tiles the U = 16 instantiation spilled 260 bytes per lane (180 with one slice out).
  extern __shared__ __attribute__((aligned(16))) unsigned char r2_raw[];
  V* const r1s = reinterpret_cast<V*>(r2_raw) + (size_t)wave_of(threadIdx.x) * (2 * U * WAVE) + (threadIdx.x & (WAVE - 1));
  V* const r2s = r1s + U * WAVE;
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t ncg = (a.n + C - 1) / C;
  V racc[U], racc2[U];
  int rgo[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int rg = wave * U + u;
    rgo[u] = (rg < a.nrg ? rg : a.nrg - 1) * (WAVE * VEC);
    V r1v, r2v;
#pragma unroll
    for (int e = 0; e < VEC; ++e) racc[u][e] = T(0), racc2[u][e] = T(0), r1v[e] = T(0), r2v[e] = T(0);
    if (rg < a.nrg) {
      r1v = *reinterpret_cast<const V*>(a.r + (int64_t)rg * (WAVE * VEC) + lane * VEC);
      r2v = *reinterpret_cast<const V*>(b.r + (int64_t)rg * (WAVE * VEC) + lane * VEC);
    }
    r1s[u * WAVE] = r1v;
    r2s[u * WAVE] = r2v;
  }
  double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};

  struct Tile {
    V col[C][U];
    T xs[C], xs2[C];
  };
  auto load = [&](Tile& t, int64_t cg) __attribute__((always_inline)) {
    const int64_t j0 = cg * C;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int64_t jc = (j0 + c < a.n) ? (j0 + c) : (a.n - 1);
      t.xs[c] = a.x[jc];
      t.xs2[c] = b.x[jc];
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int64_t jc = (j0 + c < a.n) ? (j0 + c) : (a.n - 1);
      const T* __restrict__ p = a.A + jc * a.ld;
#pragma unroll
      for (int u = 0; u < U; ++u) t.col[c][u] = nt_load(reinterpret_cast<const V*>(p + rgo[u]) + lane);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // forward-backward step of one instance for column j: y = x - gamma g ; z = prox(y) ; res = x - z ; v = z | res
  auto epilogue = [&](T g, T xj, int64_t j, bool valid, bool writer, T* g_out, T* y, T* z_new, T* res, double* ac) __attribute__((always_inline)) -> T {
    const T yj = xj - a.gamma * g;
    T zj;
    if (a.g_kind == PG_G_NORML1) {
      T th = a.p0;
      if (a.p0v != nullptr) th = pg_l1w_threshold(a.gamma, a.p0v[valid ? j : a.n - 1]);
      zj = yj <= -th ? yj + th : (yj >= th ? yj - th : T(0));
    } else if (a.g_kind == PG_G_INDBOX) {
      T lo = a.p0, hi = a.p1;
      if (a.p0v != nullptr) lo = a.p0v[valid ? j : a.n - 1], hi = a.p1v[valid ? j : a.n - 1];
      zj = fmin(hi, fmax(lo, yj));
    } else
      zj = yj;
    const T rj = xj - zj;
    if (writer && valid) {
      g_out[j] = g;
      y[j] = yj;
      z_new[j] = zj;
      res[j] = rj;
      if (a.g_kind == PG_G_NORML1) ac[0] += a.p0v != nullptr ? (double)a.p0v[j] * fabs((double)zj) : fabs((double)zj);
      ac[1] = pg_maxn(ac[1], fabs((double)rj));
      ac[2] += (double)g * (double)rj;
      ac[3] += (double)rj * (double)rj;
    }
    return valid ? (a.v_is_res ? rj : zj) : T(0);
  };
  auto process = [&](const Tile& t, int64_t cg, int buf) __attribute__((always_inline)) {
    const int64_t j0 = cg * C;
    T dot[C], dot2[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      T d = T(0), d2 = T(0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const V rv = r1s[u * WAVE];
#pragma unroll
        for (int e = 0; e < VEC; ++e) d = fma(t.col[c][u][e], rv[e], d);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const V rv = r2s[u * WAVE];
#pragma unroll
        for (int e = 0; e < VEC; ++e) d2 = fma(t.col[c][u][e], rv[e], d2);
      }
      dot[c] = wave_allsum(d);
      dot2[c] = wave_allsum(d2);
    }
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < C; ++c) sm_dot[buf][0][c][wave] = dot[c], sm_dot[buf][1][c][wave] = dot2[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < C; ++c) {
      T g = sm_dot[buf][0][c][0], g2 = sm_dot[buf][1][c][0];
#pragma unroll
      for (int w = 1; w < WAVES; ++w) g += sm_dot[buf][0][c][w], g2 += sm_dot[buf][1][c][w];
      const int64_t j = j0 + c;
      const bool valid = j < a.n;
      const bool writer = (int)threadIdx.x == c;
      const T vj = epilogue(g, t.xs[c], j, valid, writer, a.g_out, a.y, a.z_new, a.res, acc);
      const T vj2 = epilogue(g2, t.xs2[c], j, valid, writer, b.g_out, b.y, b.z_new, b.res, acc + 4);
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) racc[u][e] = fma(t.col[c][u][e], vj, racc[u][e]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) racc2[u][e] = fma(t.col[c][u][e], vj2, racc2[u][e]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) asm volatile("" : "+v"(racc[u]), "+v"(racc2[u]));
  };

  const CgMap map(ncg, C, a.line_cols, blockIdx.x, gridDim.x);
  const int64_t cnt = map.cnt;
  auto at = [&](int64_t i) { return map.at(i); };
  if constexpr (NT == 2) {
    Tile ta, tb;
    int64_t i = 0;
    if (i < cnt) load(ta, at(i));
    while (i < cnt) {
      if (i + 1 < cnt) load(tb, at(i + 1));
      process(ta, at(i), 0);
      if (i + 1 >= cnt) break;
      if (i + 2 < cnt) load(ta, at(i + 2));
      process(tb, at(i + 1), 1);
      i += 2;
    }
  } else {
    Tile t;
    int buf = 0;
    for (int64_t i = 0; i < cnt; ++i) {
      load(t, at(i));
      process(t, at(i), buf);
      buf ^= 1;
    }
  }
  T* part = a.partials + (int64_t)blockIdx.x * a.ld + lane * VEC;
  T* part2 = b.partials + (int64_t)blockIdx.x * a.ld + lane * VEC;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int rg = wave * U + u;
    if (rg < a.nrg) {
      *reinterpret_cast<V*>(part + (int64_t)rg * (WAVE * VEC)) = racc[u];
      *reinterpret_cast<V*>(part2 + (int64_t)rg * (WAVE * VEC)) = racc2[u];
    }
  }
  const double ps[8] = {a.gscale, 1.0, 1.0, 1.0, a.gscale, 1.0, 1.0, 1.0};
  grid_reduce_finalize<8, 0x22u, WAVES>(acc, a.red_partials, a.red_counter, a.scal_out, ps);
}

template <typename T, int U, int C, int WAVES, int NT>
pg_status launch_tnm_pair(pg_mat* A, TNArgs<T>& a, TNPair<T>& b, int* blocks_out) {
  pg_ctx* c = A->ctx;
  const int64_t ncg = (A->n + C - 1) / C;
  int64_t blocks = c->num_cu;
  if (blocks > ncg) blocks = ncg;
  if (blocks < 1) blocks = 1;
  PG_TRY(ensure_partials(A, (int)(2 * blocks)));
  a.partials = (T*)A->partials;
  b.partials = (T*)A->partials + blocks * A->ld;
  *blocks_out = (int)blocks;
  pg_prof_scope prof(c, PG_K_GEMV_TN);
  const size_t lds = (size_t)2 * WAVES * U * 1024;  // both instances' r slices
  if (lds + 4096 > 64 * 1024) {
    static std::mutex mu;
    static bool opted_in[64] = {};
    std::lock_guard<std::mutex> lock(mu);
    const int dev = c->device & 63;
    if (!opted_in[dev]) {
      PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_tnm_pair_kernel<T, U, C, WAVES, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      opted_in[dev] = true;
    }
  }
  hipLaunchKernelGGL((gemv_tnm_pair_kernel<T, U, C, WAVES, NT>), dim3((unsigned)blocks), dim3(WAVES * 64), lds, c->stream, a, b);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

template <typename T, int U, int C, int WAVES, int NT>
pg_status launch_tnm(pg_mat* A, TNArgs<T>& a, int* blocks_out, int bpc) {
  pg_ctx* c = A->ctx;
  const int64_t ncg = (A->n + C - 1) / C;
  int64_t blocks = (int64_t)c->num_cu * bpc;
  if (env_int("PG_TN_BLOCKS", 0) > 0) blocks = env_int("PG_TN_BLOCKS", 0);
  if (blocks > ncg) blocks = ncg;
  if (blocks > PG_RED_MAX_BLOCKS) blocks = PG_RED_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  PG_TRY(ensure_partials(A, (int)blocks));
  a.partials = (T*)A->partials;
  *blocks_out = (int)blocks;
  pg_prof_scope prof(c, PG_K_GEMV_TN);
  hipLaunchKernelGGL((gemv_tnm_kernel<T, U, C, WAVES, NT>), dim3((unsigned)blocks), dim3(WAVES * 64), 0, c->stream, a);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

}  // namespace

// The instantiations: U row groups per wave, C columns per step, W waves, NT register tiles.
//   29 .. 32 row groups   eight waves of U = 4, C = 4, two tiles           (8192 x 2^18: 7.16-7.20 TB/s against 7.04-7.08)
//   33 .. 64              four waves of U = ceil(nrg / 4), C = 2, two tiles (16384 x 2^20, the headline: 0.914-0.917 of 8 TB/s
//                         against 0.894-0.896 for gemv_tn_kernel<16, 2, 4>); U = 10: C = 4, one tile
//   65 .. 128             eight waves of U = ceil(nrg / 8), one tile, C = 2 up to U = 12, C = 1 above
// (interleaved A/B on one box, medians of five rounds: profiles/r3_tune_tn_mid_columns.log)
template <typename T>
pg_status launch_tn_mid(pg_mat* A, TNArgs<T>& a, int* blocks_out, int U, int C, int W, int nt, int bpc) {
#define PG_TNM(UU, CC, WW, NN) \
  if (U == UU && C == CC && W == WW && nt == NN) return launch_tnm<T, UU, CC, WW, NN>(A, a, blocks_out, bpc)
  PG_TNM(4, 4, 8, 2);
  PG_TNM(9, 2, 4, 2); PG_TNM(10, 2, 4, 2); PG_TNM(11, 2, 4, 2); PG_TNM(12, 2, 4, 2);
  PG_TNM(13, 2, 4, 2); PG_TNM(14, 2, 4, 2); PG_TNM(15, 2, 4, 2); PG_TNM(16, 2, 4, 2);
  PG_TNM(10, 4, 4, 1); PG_TNM(16, 2, 4, 1);
  PG_TNM(9, 2, 8, 1); PG_TNM(10, 2, 8, 1); PG_TNM(11, 2, 8, 1); PG_TNM(12, 2, 8, 1);
  PG_TNM(13, 1, 8, 1); PG_TNM(14, 1, 8, 1); PG_TNM(15, 1, 8, 1); PG_TNM(16, 1, 8, 1);
#undef PG_TNM
  pg_set_error("no gemv_tnm instantiation for U=%d C=%d WAVES=%d tiles=%d", U, C, W, nt);
  return PG_ERR_UNSUPPORTED;
}

bool tn_mid_covers(int nrg) { return nrg >= 29 && nrg <= 128; }

// The pair sweep: columns of 33 .. 64 row groups (config 4's 16384 rows among them); both instances' r slices in LDS (128 KiB at
// 64 row groups), both image accumulator sets and the tile in registers.
bool tn_pair_covers(int nrg) { return nrg >= 33 && nrg <= 64; }

template <typename T>
pg_status launch_tn_pair(pg_mat* A, TNArgs<T>& a, const T* r2, const T* x2, T* g2, T* y2, T* z2, T* res2, int* blocks_out, T** partials2_out) {
  if (!tn_pair_covers(a.nrg)) {
    pg_set_error("the two-point sweep covers columns of %d .. %d rows", (int)(33 * 1024 / sizeof(T)) - (int)(1024 / sizeof(T)) + 1, (int)(64 * 1024 / sizeof(T)));
    return PG_ERR_UNSUPPORTED;
  }
  TNPair<T> b;
  b.r = r2, b.x = x2, b.g_out = g2, b.y = y2, b.z_new = z2, b.res = res2, b.partials = nullptr;
  // Eight waves of U = ceil(nrg / 8) row groups, TWO columns per step, ONE register tile at U = 7, 8 (two waves per SIMD cover each other's
  // loads): measured at config 4's size (scripts/r5_pair_sweep_rate.py, profiles/r5_pair_sweep_rate.log) 9.26 ms against the single
  // sweep's 9.09 -- both instances for 1.02 single sweeps, 0.885 of 8 TB/s -- where four waves of U = 16, one column per step and
  // two tiles take 10.66 ms (1.17), eight waves with one column and two tiles 9.75 (1.07), four waves, two columns, one tile 9.62.
  const int U = (a.nrg + 7) / 8;
  pg_status st = PG_ERR_UNSUPPORTED;
  // U <= 6 (33 .. 48 row groups) leaves the registers for a second tile (215 / 245 registers in Float32, 225 / 256 in Float64, nothing in
  // scratch): 9216 rows 1.17 -> 1.08 single sweeps, 10240 rows 1.10 -> 1.03, 12288 rows 1.06 -> 1.02; at U = 7 it spills and costs 14 %
  // (profiles/r5_pair_sweep_rate.log, seventh collection)
  const int nt = U <= 6 ? 2 : 1;
#define PG_TNMP(UU, NN) \
  if (U == UU && nt == NN) st = launch_tnm_pair<T, UU, 2, 8, NN>(A, a, b, blocks_out)
  PG_TNMP(5, 2); PG_TNMP(6, 2); PG_TNMP(7, 1); PG_TNMP(8, 1);
#undef PG_TNMP
  *partials2_out = b.partials;
  return st;
}
template pg_status launch_tn_pair<float>(pg_mat*, TNArgs<float>&, const float*, const float*, float*, float*, float*, float*, int*, float**);
template pg_status launch_tn_pair<double>(pg_mat*, TNArgs<double>&, const double*, const double*, double*, double*, double*, double*, int*, double**);


// geometry by column length (see the table above)
template <typename T>
pg_status launch_tn_mid_default(pg_mat* A, TNArgs<T>& a, int* blocks_out) {
  const int nrg = a.nrg;
  if (nrg <= 32) return launch_tn_mid<T>(A, a, blocks_out, 4, 4, 8, 2, 1);
  if (nrg <= 64) {
    const int U = (nrg + 3) / 4;
    return U == 10 ? launch_tn_mid<T>(A, a, blocks_out, 10, 4, 4, 1, 1) : launch_tn_mid<T>(A, a, blocks_out, U, 2, 4, 2, 1);
  }
  const int U = (nrg + 7) / 8;
  return launch_tn_mid<T>(A, a, blocks_out, U, U <= 12 ? 2 : 1, 8, 1, 1);
}

template pg_status launch_tn_mid<float>(pg_mat*, TNArgs<float>&, int*, int, int, int, int, int);
template pg_status launch_tn_mid<double>(pg_mat*, TNArgs<double>&, int*, int, int, int, int, int);
template pg_status launch_tn_mid_default<float>(pg_mat*, TNArgs<float>&, int*);
template pg_status launch_tn_mid_default<double>(pg_mat*, TNArgs<double>&, int*);

}  // namespace pgtn
