// Columns of 29 .. 128 row groups (7424 .. 32768 rows in Float32, the headline's 16384 among them): gemv_tnm_kernel, the
// one-workgroup sweep with a row-group count per wave U that fits the column exactly (9 .. 16 rather than the next power
// of two), branch-free tile loads and x_j / z_old_j issued ahead of the tile, compiled in its own translation unit.
// Measurements behind it: scripts/tile_pattern.hip and scripts/r3_mid_sweep.py (profiles/r3_mid_columns_counters.md,
// profiles/r3_tune_tn_mid_columns.log).
#include "pg_gemv_tn.h"

namespace pgtn {

namespace {

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
        acc[1] = fmax(acc[1], fabs((double)rj));
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
