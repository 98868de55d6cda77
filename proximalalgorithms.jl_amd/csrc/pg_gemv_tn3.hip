// Mid-length columns (33 .. 128 row groups: 8448 .. 32768 rows in Float32): instantiations of the one-workgroup sweep
// (gemv_tn_kernel, pg_gemv_tn.h) for a row-group count per wave U that fits the column exactly (9 .. 16 rather than the
// next power of two) and with the register-tile count chosen per geometry, compiled in their own translation unit.
// scripts/tile_pattern.hip (profiles/r3_mid_columns_counters.md) is the measurement behind them: the load pattern of these
// geometries alone reaches 7.2-7.35 TB/s with the dot / barrier / accumulate structure of the sweep, so what the
// power-of-two geometries lose at these lengths is repeated or wasted loads, not the memory system.
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
//   * x_j and z_old_j were vector loads issued AFTER the tile had arrived (s_waitcnt vmcnt(0) in front of them), so every
//     step paid one more memory round trip between its barrier and its epilogue with nothing in flight.  Here they are
//     SCALAR loads (constant address space: s_load_dword, its own counter, no in-order queue shared with the tile) issued
//     before the tile's loads.
// NT = 1: one register tile (load, wait, consume); NT = 2: the next tile's loads are in flight while this one is consumed.
// ---------------------------------------------------------------------------------------------------------------
// x_j / z_old_j: one scalar load each, issued where the source says (volatile asm; the compiler's own s_load was sunk to
// the instruction in front of the barrier) and waited for with sload_wait right before the epilogue
__device__ __forceinline__ unsigned sload_bits(const float* p) {
  unsigned v;
  asm volatile("s_load_dword %0, %1, 0x0" : "=s"(v) : "s"(p));
  return v;
}
__device__ __forceinline__ unsigned long long sload_bits(const double* p) {
  unsigned long long v;
  asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=s"(v) : "s"(p));
  return v;
}
__device__ __forceinline__ float sload_wait(unsigned& b) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b));
  return __builtin_bit_cast(float, b);
}
__device__ __forceinline__ double sload_wait(unsigned long long& b) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b));
  return __builtin_bit_cast(double, b);
}
template <typename T>
struct BitsOf;
template <>
struct BitsOf<float> {
  using type = unsigned;
};
template <>
struct BitsOf<double> {
  using type = unsigned long long;
};

template <typename T, int U, int C, int WAVES, int NT>
__global__ __launch_bounds__(WAVES * 64) void gemv_tnm_kernel(TNArgs<T> a) {
  using V = typename VecOf<T>::type;
  constexpr int VEC = VecOf<T>::N;
  __shared__ T sm_dot[2][C][WAVES];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t ncg = (a.n + C - 1) / C;
  using Bits = typename BitsOf<T>::type;

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
    Bits xs[C], zos[C];
  };
  auto load = [&](Tile& t, int64_t cg) __attribute__((always_inline)) {
    const int64_t j0 = cg * C;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int64_t jc = (j0 + c < a.n) ? (j0 + c) : (a.n - 1);
      t.xs[c] = sload_bits(a.x + jc);
      t.zos[c] = sload_bits(a.z_old + jc);
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
  auto process = [&](Tile& t, int64_t cg, int buf) __attribute__((always_inline)) {
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
      const T xj = sload_wait(t.xs[c]), zo = sload_wait(t.zos[c]);
      const T yj = xj - a.gamma * g;  // forward_backward.jl:117 / fast_forward_backward.jl:140
      T zj;                            // :118 / :141
      if (a.g_kind == PG_G_NORML1)
        zj = yj <= -a.p0 ? yj + a.p0 : (yj >= a.p0 ? yj - a.p0 : T(0));
      else if (a.g_kind == PG_G_INDBOX)
        zj = fmin(a.p1, fmax(a.p0, yj));
      else
        zj = yj;
      const T rj = xj - zj;                                   // :120 / :142
      const T vj = valid ? zj + a.beta * (zj - zo) : T(0);    // fast_forward_backward.jl:135 of the next iteration
      if ((int)threadIdx.x == c && valid) {
        a.g_out[j] = g;
        a.y[j] = yj;
        a.z_new[j] = zj;
        a.res[j] = rj;
        if (a.v_out != nullptr) a.v_out[j] = vj;
        if (a.g_kind == PG_G_NORML1) acc[0] += fabs((double)zj);
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

  const int64_t cnt = ncg > (int64_t)blockIdx.x ? (ncg - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  auto at = [&](int64_t i) { return (int64_t)blockIdx.x + i * (int64_t)gridDim.x; };
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

template <typename T, int U, int C, int WAVES, bool DB>
pg_status launch_mid(pg_mat* A, TNArgs<T>& a, int* blocks_out, int bpc) {
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
  hipLaunchKernelGGL((gemv_tn_kernel<T, U, C, WAVES, DB, 0>), dim3((unsigned)blocks), dim3(WAVES * 64), 0, c->stream, a);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

}  // namespace

// U row groups per wave, C columns per step, W waves, db register tiles - 1; PG_ERR_UNSUPPORTED when not instantiated
template <typename T>
pg_status launch_tn_mid(pg_mat* A, TNArgs<T>& a, int* blocks_out, int U, int C, int W, int db, int bpc) {
  if (env_int("PG_TN_MIDK", 1) == 1) {
#define PG_TNM(UU, CC, WW, NN) \
  if (U == UU && C == CC && W == WW && db + 1 == NN) return launch_tnm<T, UU, CC, WW, NN>(A, a, blocks_out, bpc)
    PG_TNM(5, 4, 4, 1); PG_TNM(6, 4, 4, 1); PG_TNM(7, 4, 4, 1); PG_TNM(8, 4, 4, 1);
    PG_TNM(5, 4, 4, 2); PG_TNM(6, 4, 4, 2); PG_TNM(7, 4, 4, 2); PG_TNM(8, 4, 4, 2);
    PG_TNM(8, 2, 4, 1); PG_TNM(8, 2, 4, 2);
    PG_TNM(4, 8, 8, 1); PG_TNM(4, 4, 8, 1); PG_TNM(4, 4, 8, 2);
    PG_TNM(9, 2, 4, 1); PG_TNM(10, 2, 4, 1); PG_TNM(11, 2, 4, 1); PG_TNM(12, 2, 4, 1);
    PG_TNM(13, 2, 4, 1); PG_TNM(14, 2, 4, 1); PG_TNM(15, 2, 4, 1); PG_TNM(16, 2, 4, 1);
    PG_TNM(9, 2, 4, 2); PG_TNM(10, 2, 4, 2); PG_TNM(11, 2, 4, 2); PG_TNM(12, 2, 4, 2);
    PG_TNM(13, 2, 4, 2); PG_TNM(14, 2, 4, 2); PG_TNM(15, 2, 4, 2); PG_TNM(16, 2, 4, 2);
    PG_TNM(10, 4, 4, 1); PG_TNM(12, 4, 4, 1);
    PG_TNM(9, 1, 8, 1); PG_TNM(10, 1, 8, 1); PG_TNM(11, 1, 8, 1); PG_TNM(12, 1, 8, 1);
    PG_TNM(13, 1, 8, 1); PG_TNM(14, 1, 8, 1); PG_TNM(15, 1, 8, 1); PG_TNM(16, 1, 8, 1);
    PG_TNM(9, 2, 8, 1); PG_TNM(10, 2, 8, 1); PG_TNM(11, 2, 8, 1); PG_TNM(12, 2, 8, 1); PG_TNM(13, 2, 8, 1);
    PG_TNM(9, 1, 8, 2); PG_TNM(10, 1, 8, 2);
#undef PG_TNM
    pg_set_error("no gemv_tnm instantiation for U=%d C=%d WAVES=%d tiles=%d", U, C, W, db + 1);
    return PG_ERR_UNSUPPORTED;
  }
#define PG_MID(UU, CC, WW, DD) \
  if (U == UU && C == CC && W == WW && db == DD) return launch_mid<T, UU, CC, WW, (DD != 0)>(A, a, blocks_out, bpc)
  // four waves, exact U (33 .. 64 row groups)
  PG_MID(9, 2, 4, 1); PG_MID(10, 2, 4, 1); PG_MID(11, 2, 4, 1); PG_MID(12, 2, 4, 1);
  PG_MID(13, 2, 4, 1); PG_MID(14, 2, 4, 1); PG_MID(15, 2, 4, 1);
  PG_MID(9, 2, 4, 0); PG_MID(10, 2, 4, 0); PG_MID(11, 2, 4, 0); PG_MID(12, 2, 4, 0);
  PG_MID(13, 2, 4, 0); PG_MID(14, 2, 4, 0); PG_MID(15, 2, 4, 0); PG_MID(16, 2, 4, 0);
  PG_MID(10, 4, 4, 0); PG_MID(12, 4, 4, 0);
  // 17 .. 32 row groups on four waves (U = 5 .. 8)
  PG_MID(8, 4, 4, 0); PG_MID(8, 2, 4, 0); PG_MID(8, 4, 4, 1); PG_MID(8, 2, 4, 1);
  PG_MID(5, 4, 4, 0); PG_MID(6, 4, 4, 0); PG_MID(7, 4, 4, 0);
  PG_MID(5, 4, 4, 1); PG_MID(6, 4, 4, 1); PG_MID(7, 4, 4, 1);
  // eight waves, exact U (65 .. 128 row groups)
  PG_MID(9, 2, 8, 0); PG_MID(10, 2, 8, 0); PG_MID(11, 2, 8, 0); PG_MID(12, 2, 8, 0);
  PG_MID(13, 2, 8, 0); PG_MID(14, 2, 8, 0); PG_MID(15, 2, 8, 0); PG_MID(16, 2, 8, 0);
  PG_MID(9, 1, 8, 0); PG_MID(10, 1, 8, 0); PG_MID(11, 1, 8, 0); PG_MID(12, 1, 8, 0);
  PG_MID(13, 1, 8, 0); PG_MID(14, 1, 8, 0); PG_MID(15, 1, 8, 0);
  PG_MID(9, 1, 8, 1); PG_MID(12, 1, 8, 1); PG_MID(16, 1, 8, 1);
#undef PG_MID
  pg_set_error("no mid-column gemv_tn instantiation for U=%d C=%d WAVES=%d tiles=%d", U, C, W, db + 1);
  return PG_ERR_UNSUPPORTED;
}

template pg_status launch_tn_mid<float>(pg_mat*, TNArgs<float>&, int*, int, int, int, int, int);
template pg_status launch_tn_mid<double>(pg_mat*, TNArgs<double>&, int*, int, int, int, int, int);

}  // namespace pgtn
