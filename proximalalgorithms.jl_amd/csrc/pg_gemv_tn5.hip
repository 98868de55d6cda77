// The three-point sweep: THREE instances of the matrix-level single sweep (pg_mat_fused_tn) on ONE read of A, for the trial
// points tau, tau / 2 and tau / 4 of ZeroFPR's line search (zerofpr.jl:200-217).  At config 4's instance 10 of the first 23 line
// searches accept tau = 1/4: with two points per sweep (gemv_tnm_pair_kernel, pg_gemv_tn3.hip) each of them read A a second time.
//
// What one CU can hold decides the shape.  At 64 row groups (16384 rows in Float32) an instance's slice of r and its image
// accumulators are 64 KiB each and a two-column tile is 128 KiB: three instances are 512 KiB of state on a CU with 512 KiB of
// registers and 160 KiB of LDS.  The slices of instances 0 and 1 live in LDS (128 KiB), the third's in registers except its first
// R3L row groups per wave (LDS again: 8 R3L KiB), and the twelve fp64 scalar accumulators, which only the writer lanes of wave 0
// touch, in LDS as well (in registers they cost every lane 24 and the kernel spilled); the three image accumulator sets and
// the tile in registers.  Per column and instance the arithmetic is gemv_tnm_pair_kernel's except that a lane's part of a dot runs as
// two partial sums (even / odd elements, v_pk_fma_f32 in Float32): results equal the single sweep's to rounding, like the pair's.
#include <mutex>

#include "pg_gemv_tn.h"

namespace pgtn {

namespace {

template <typename T>
struct TNTrio {
  const T* r[3];  // [ld] each
  const T* x[3];  // [n] each
  T *g_out[3], *y[3], *z_new[3], *res[3];  // [n] each
  T* partials[3];                          // [gridDim.x][ld] each
};

template <typename T, int U, int C, int WAVES, int R3L, int NT>
__global__ __launch_bounds__(WAVES * 64) void gemv_tnm_trio_kernel(TNArgs<T> a, TNTrio<T> b) {
  using V = typename VecOf<T>::type;
  constexpr int VEC = VecOf<T>::N;
  typedef T T2 __attribute__((ext_vector_type(2)));
  constexpr int UL = 2 * U + R3L;  // row groups of LDS per wave
  __shared__ __attribute__((aligned(16))) T sm_dot[2][3][C][WAVES];
  __shared__ double sm_acc[3 * C][4];
  extern __shared__ __attribute__((aligned(16))) unsigned char r_raw[];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  V* const r1s = reinterpret_cast<V*>(r_raw) + (size_t)wave * (UL * WAVE) + lane;
  V* const r2s = r1s + U * WAVE;
  V* const r3s = r2s + U * WAVE;
  const int64_t ncg = (a.n + C - 1) / C;
  V racc[3][U], r3[U - R3L > 0 ? U - R3L : 1];  // (R3L = U: the third slice whole in LDS)
  int rgo[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int rg = wave * U + u;
    rgo[u] = (rg < a.nrg ? rg : a.nrg - 1) * (WAVE * VEC);
    V rv[3];
#pragma unroll
    for (int e = 0; e < VEC; ++e) rv[0][e] = T(0), rv[1][e] = T(0), rv[2][e] = T(0);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) racc[k][u][e] = T(0);
    }
    if (rg < a.nrg) {
#pragma unroll
      for (int p = 0; p < 3; ++p) rv[p] = *reinterpret_cast<const V*>(b.r[p] + (int64_t)rg * (WAVE * VEC) + lane * VEC);
    }
    r1s[u * WAVE] = rv[0];
    r2s[u * WAVE] = rv[1];
    if (u < R3L) r3s[u * WAVE] = rv[2];
    else r3[u < R3L ? 0 : u - R3L] = rv[2];
  }
  // The epilogues of a step's 3 C (instance, column) pairs run in PARALLEL LANES, pair q = p C + c in lane q of every wave (its own
  // sum of the waves' partial dots, its own x_j, prox and residual; lane q of wave 0 writes the pair's outputs and keeps its four
  // scalars), and the 3 C values v come back to all lanes by v_readlane.  One after the other in every lane they were six
  // branchy chains of an LDS read, seven adds and the prox between the barrier and the image fmas of every step, with nothing of
  // this workgroup in flight: 9.92 ms per sweep at config 4's size against 9.28 this way (profiles/r5_pair_sweep_rate.log).
  const int ql = lane % (3 * C), pl = ql / C, cl = ql % C;
  const T* const xb = pl == 0 ? b.x[0] : (pl == 1 ? b.x[1] : b.x[2]);
  T* const gob = pl == 0 ? b.g_out[0] : (pl == 1 ? b.g_out[1] : b.g_out[2]);
  T* const yb = pl == 0 ? b.y[0] : (pl == 1 ? b.y[1] : b.y[2]);
  T* const zb = pl == 0 ? b.z_new[0] : (pl == 1 ? b.z_new[1] : b.z_new[2]);
  T* const rb = pl == 0 ? b.res[0] : (pl == 1 ? b.res[1] : b.res[2]);
  const bool owner = wave == 0 && lane < 3 * C;
  if (owner) {
#pragma unroll
    for (int k = 0; k < 4; ++k) sm_acc[lane][k] = 0.0;
  }

  struct Tile {
    V col[C][U];
    T xl;  // x_j of this lane's (instance, column) pair
  };
  auto load = [&](Tile& t, int64_t cg) __attribute__((always_inline)) {
    const int64_t j0 = cg * C;
    __builtin_amdgcn_sched_barrier(0);
    t.xl = xb[(j0 + cl < a.n) ? (j0 + cl) : (a.n - 1)];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int64_t jc = (j0 + c < a.n) ? (j0 + c) : (a.n - 1);
      const T* __restrict__ p = a.A + jc * a.ld;
#pragma unroll
      for (int u = 0; u < U; ++u) t.col[c][u] = nt_load(reinterpret_cast<const V*>(p + rgo[u]) + lane);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto process = [&](const Tile& t, int64_t cg, int buf) __attribute__((always_inline)) {
    const int64_t j0 = cg * C;
    T dot[3][C];
    // two partial sums per instance, column and lane (even / odd elements): Float32 runs them as v_pk_fma_f32, half the issue
    // slots of the single sweep's one chain -- with three instances on a tile the dots are what the step waits for.  Row groups
    // outermost: a slice entry is read from LDS once for the C columns of the step.
    T2 dd[3][C];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int c = 0; c < C; ++c) dd[p][c] = T2{T(0), T(0)};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      V rv[3];
      rv[0] = r1s[u * WAVE];
      rv[1] = r2s[u * WAVE];
      if constexpr (R3L > 0) rv[2] = u < R3L ? r3s[u * WAVE] : r3[u < R3L ? 0 : u - R3L];
      else rv[2] = r3[u];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
#pragma unroll
          for (int e = 0; e < VEC; e += 2) dd[p][c] = T2{t.col[c][u][e], t.col[c][u][e + 1]} * T2{rv[p][e], rv[p][e + 1]} + dd[p][c];
        }
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
      for (int p = 0; p < 3; ++p) dot[p][c] = wave_allsum(dd[p][c][0] + dd[p][c][1]);
    }
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int p = 0; p < 3; ++p) sm_dot[buf][p][c][wave] = dot[p][c];
      }
    }
    __syncthreads();
    // forward-backward step of this lane's pair (gemv_tnm_pair_kernel's statements); the scalars in LDS
    const T* sp = &sm_dot[buf][pl][cl][0];
    T g = sp[0];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) g += sp[w];
    const int64_t j = j0 + cl;
    const bool valid = j < a.n;
    const T xj = t.xl;
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
    if (owner && valid) {
      gob[j] = g;
      yb[j] = yj;
      zb[j] = zj;
      rb[j] = rj;
      double* ac = &sm_acc[lane][0];
      if (a.g_kind == PG_G_NORML1) ac[0] += a.p0v != nullptr ? (double)a.p0v[j] * fabs((double)zj) : fabs((double)zj);
      ac[1] = pg_maxn(ac[1], fabs((double)rj));
      ac[2] += (double)g * (double)rj;
      ac[3] += (double)rj * (double)rj;
    }
    const T vl = valid ? (a.v_is_res ? rj : zj) : T(0);
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const T vk = pg_readlane(vl, k * C + c);
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) racc[k][u][e] = fma(t.col[c][u][e], vk, racc[k][u][e]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
      for (int u = 0; u < U; ++u) asm volatile("" : "+v"(racc[k][u]));
    }
  };

  const CgMap map(ncg, C, a.line_cols, blockIdx.x, gridDim.x);
  const int64_t cnt = map.cnt;
  if constexpr (NT == 2) {  // shorter columns leave the registers for it: the next tile's loads in flight while this one is consumed
    Tile ta, tb;
    int64_t i = 0;
    if (i < cnt) load(ta, map.at(i));
    while (i < cnt) {
      if (i + 1 < cnt) load(tb, map.at(i + 1));
      process(ta, map.at(i), 0);
      if (i + 1 >= cnt) break;
      if (i + 2 < cnt) load(ta, map.at(i + 2));
      process(tb, map.at(i + 1), 1);
      i += 2;
    }
  } else {
    Tile t;
    int buf = 0;
    for (int64_t i = 0; i < cnt; ++i) {
      load(t, map.at(i));
      process(t, map.at(i), buf);
      buf ^= 1;
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    T* part = b.partials[k] + (int64_t)blockIdx.x * a.ld + lane * VEC;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rg = wave * U + u;
      if (rg < a.nrg) *reinterpret_cast<V*>(part + (int64_t)rg * (WAVE * VEC)) = racc[k][u];
    }
  }
  double acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = (owner && pl == k / 4) ? sm_acc[owner ? lane : 0][k % 4] : 0.0;
  const double ps[12] = {a.gscale, 1.0, 1.0, 1.0, a.gscale, 1.0, 1.0, 1.0, a.gscale, 1.0, 1.0, 1.0};
  grid_reduce_finalize<12, 0x222u, WAVES>(acc, a.red_partials, a.red_counter, a.scal_out, ps);
}

template <typename T, int U, int C, int WAVES, int R3L, int NT>
pg_status launch_tnm_trio(pg_mat* A, TNArgs<T>& a, TNTrio<T>& b, int* blocks_out) {
  pg_ctx* c = A->ctx;
  const int64_t ncg = (A->n + C - 1) / C;
  int64_t blocks = c->num_cu;
  if (blocks > ncg) blocks = ncg;
  if (blocks < 1) blocks = 1;
  PG_TRY(ensure_partials(A, (int)(3 * blocks)));
  for (int k = 0; k < 3; ++k) b.partials[k] = (T*)A->partials + (int64_t)k * blocks * A->ld;
  a.partials = b.partials[0];
  *blocks_out = (int)blocks;
  pg_prof_scope prof(c, PG_K_GEMV_TN);
  const size_t lds = (size_t)WAVES * (2 * U + R3L) * 1024;
  if (lds + 4096 > 64 * 1024) {
    static std::mutex mu;
    static bool opted_in[64] = {};
    std::lock_guard<std::mutex> lock(mu);
    const int dev = c->device & 63;
    if (!opted_in[dev]) {
      PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_tnm_trio_kernel<T, U, C, WAVES, R3L, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      opted_in[dev] = true;
    }
  }
  hipLaunchKernelGGL((gemv_tnm_trio_kernel<T, U, C, WAVES, R3L, NT>), dim3((unsigned)blocks), dim3(WAVES * 64), lds, c->stream, a, b);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

template <typename T>
pg_status launch_tn_trio(pg_mat* A, TNArgs<T>& a, TNTrio<T>& b, int* blocks_out) {
  // eight waves of U = ceil(nrg / 8) row groups, two columns per step, one register tile (two at U = 5): the pair sweep's geometry.  At U = 8
  // the third slice keeps three of its eight row groups per wave in LDS (152 KiB with the two whole slices): 238 registers in
  // Float32, 252 in Float64, nothing in scratch (all of it in registers: 228 / 240 bytes per lane spilled).
  const int U = (a.nrg + 7) / 8;
  // U = 5 (33 .. 40 row groups) leaves the registers for a second tile -- 244 / 248 registers, nothing in scratch; 9216 rows: 1.13 -> 1.05
  // single sweeps, 10240 rows: 1.07 -> 1.01 (profiles/r5_pair_sweep_rate.log); at U = 6 it spills 32 / 56 bytes per lane and gains 0.5 %: one tile
  const int nt = U == 5 ? 2 : 1;
#define PG_TNMT(UU, LL, NN) \
  if (U == UU && nt == NN) return launch_tnm_trio<T, UU, 2, 8, LL, NN>(A, a, b, blocks_out)
  PG_TNMT(5, 0, 2); PG_TNMT(6, 0, 1); PG_TNMT(7, 0, 1); PG_TNMT(8, 3, 1);
#undef PG_TNMT
  pg_set_error("no three-point sweep for %d row groups", a.nrg);
  return PG_ERR_UNSUPPORTED;
}

// THREE instances of mat_fused_tn_t (pg_gemv.hip) on ONE read of A: the same gamma and g, three pairs (r, x); the vector outputs
// and the image three times; scalars -> dscal[PG_S_PAIR .. + 12).
template <typename T>
pg_status mat_fused_tn_trio_t(pg_mat* A, const void* const* r, const void* const* x, double gamma, int g_kind, double g_p0, double g_p1,
                              void* const* At_r, void* const* y, void* const* z, void* const* res, void* const* Az, bool image_of_res) {
  pg_ctx* c = A->ctx;
  const int nrg = (int)(A->ld / (1024 / (int64_t)sizeof(T)));
  // (the guard of the two-point entry point, tn_supported<T> included: the sweeps index column groups in 32 bits)
  if (pg_row_sharded(c) || pg_col_sharded(c) || A->m <= 0 || A->n <= 0 || A->n >= ((int64_t)1 << 31) || !tn_pair_covers(nrg)) {
    pg_set_error("the three-point sweep needs an unsharded operator with %d .. %d rows", (int)(32 * (1024 / sizeof(T)) + 1), (int)(64 * (1024 / sizeof(T))));
    return PG_ERR_UNSUPPORTED;
  }
  void** pads[3] = {&A->rpad, &A->rpad2, &A->rpad3};
  for (int p = 0; p < 3; ++p) {
    if (*pads[p] == nullptr) {
      PG_HIP(hipMalloc(pads[p], (size_t)A->ld * sizeof(T)));
      PG_HIP(hipMemsetAsync(*pads[p], 0, (size_t)A->ld * sizeof(T), c->stream));
    }
    PG_HIP(hipMemcpyAsync(*pads[p], r[p], (size_t)A->m * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
  }
  TNArgs<T> a;
  a.A = (const T*)A->data;
  a.ld = A->ld;
  a.n = A->n;
  a.m = A->m;
  a.nrg = nrg;
  a.r = (const T*)A->rpad;
  a.x = (const T*)x[0];
  a.z_old = a.x;
  const T gm = (T)gamma;
  a.gamma = gm;
  a.beta = T(0);
  a.v_is_res = image_of_res ? 1 : 0;
  a.p0 = g_kind == PG_G_NORML1 ? (T)(gm * (T)g_p0) : (T)g_p0;
  a.p1 = (T)g_p1;
  a.lam_ls = T(1);
  a.g_kind = g_kind;
  a.gscale = g_kind == PG_G_NORML1 ? (double)(T)g_p0 : 0.0;
  a.g_out = (T*)At_r[0];
  a.y = (T*)y[0];
  a.z_new = (T*)z[0];
  a.res = (T*)res[0];
  a.v_out = nullptr;
  a.partials = nullptr;
  a.red_partials = c->red_partials;
  a.red_counter = c->red_counter;
  a.scal_out = c->dscal + PG_S_PAIR;
  a.line_cols = 32;
  TNTrio<T> b;
  for (int p = 0; p < 3; ++p) {
    b.r[p] = (const T*)*pads[p];
    b.x[p] = (const T*)x[p];
    b.g_out[p] = (T*)At_r[p], b.y[p] = (T*)y[p], b.z_new[p] = (T*)z[p], b.res[p] = (T*)res[p];
  }
  int blocks = 0;
  PG_TRY(launch_tn_trio<T>(A, a, b, &blocks));
  int64_t fb = (A->ld + 63) / 64;
  if (fb > 1024) fb = 1024;
  pg_prof_scope prof(c, PG_K_GEMV_N_FINISH);
  for (int k = 0; k < 3; ++k) {
    hipLaunchKernelGGL((gemv_n_finish_kernel<T, false>), dim3((unsigned)fb), dim3(1024), 0, c->stream, (const T*)b.partials[k], A->ld, A->m, blocks,
                       (const T*)nullptr, (T*)Az[k], A->m, 0.0, (double*)nullptr, (unsigned*)nullptr, (double*)nullptr, (T*)nullptr, ColPack<T>{});
    PG_LAUNCH_CHECK();
  }
  return PG_OK;
}

}  // namespace
}  // namespace pgtn

extern "C" pg_status pg_mat_fused_tn_trio(pg_mat* A, const void* const r[3], const void* const x[3], double gamma, int32_t g_kind, double g_p0,
                                          double g_p1, void* const At_r[3], void* const y[3], void* const z[3], void* const res[3],
                                          void* const Az[3], int32_t image_of_res, double* scalars_out) {
  PG_REQUIRE(A != nullptr, "matrix is null");
  PG_REQUIRE(r && x && At_r && y && z && res && Az, "null argument");
  for (int p = 0; p < 3; ++p) PG_REQUIRE(r[p] && x[p] && At_r[p] && y[p] && z[p] && res[p] && Az[p], "null vector");
  PG_REQUIRE(g_kind == PG_G_ZERO || g_kind == PG_G_NORML1 || g_kind == PG_G_INDBOX, "unknown g_kind");
  PG_REQUIRE(gamma > 0, "gamma must be positive");
  PG_TRY(A->dtype == PG_F32 ? pgtn::mat_fused_tn_trio_t<float>(A, r, x, gamma, g_kind, g_p0, g_p1, At_r, y, z, res, Az, image_of_res != 0)
                            : pgtn::mat_fused_tn_trio_t<double>(A, r, x, gamma, g_kind, g_p0, g_p1, At_r, y, z, res, Az, image_of_res != 0));
  if (scalars_out) {
    PG_TRY(pg_read_scalars(A->ctx, PG_S_PAIR, 12));
    for (int k = 0; k < 12; ++k) scalars_out[k] = A->ctx->hscal[PG_S_PAIR + k];
  }
  return PG_OK;
}
