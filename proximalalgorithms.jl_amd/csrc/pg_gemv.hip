// The two GEMV orientations on the column-major store of A, and the LeastSquares operator built on
// them (reference: benchmark/benchmarks.jl:11-17  res = A*x - b ; (norm(res)^2/2, A'res)).
//
// Both passes are pure HBM streams over A (0.5 flop/byte in f32): no MFMA, no LDS staging of A.  Every
// wave issues 16-byte-per-lane loads (1 KiB per wave instruction) of whole 1 KiB row groups of a column,
// non-temporal because A (64 GiB at the headline size) is streamed exactly once per pass.
//
//   pass N  (y = A x):     wave = (row tile of R KiB, column slot); accumulators in VGPRs, x_j wave-uniform;
//                          deterministic two-stage reduction over the column slots.
//   pass T  (g = A' r):    r staged once per workgroup in LDS (<= 128 KiB); wave = C adjacent columns streamed
//                          top to bottom; per-column DPP/shuffle wave reduction; no cross-wave traffic.
//   pass TN (single sweep): g = A' r, then per finished column the prox step and the column's contribution to the
//                          NEXT residual while it is still in registers -- A read once per iteration
//                          (pg_ls_fused_pass / pg_mat_fused_tn; column shards: one all-reduce of m + 8 N elements).
//                          Three geometries by column length: gemv_tnw (short), gemv_tn (pg_gemv_tn.h), gemv_tnt
//                          (long: teams of workgroups) -- see launch_tn.
#include <mutex>

#include "pg_gemv_tn.h"

using namespace pgtn;

namespace {

// -------------------------------------------------------------------------------------------------
// pass N, stage 1: partials[slot][i] = sum_{j in columns of slot} A[i, j] * x[j]
// Workgroup = 4 waves = TB adjacent row tiles x TW column slots (TB * TW = 4).  Waves that share a slot take
// adjacent row tiles, so they read TB*R KiB contiguous bytes of each column; waves that share a tile are
// combined through LDS (fixed order) before the partial is written: S_eff = ceil(S / TW) partial vectors.
// -------------------------------------------------------------------------------------------------
template <typename T, int R, int U, int TW>
__global__ __launch_bounds__(256) void gemv_n_partial_kernel(const T* __restrict__ A, int64_t ld, int64_t n,
                                                             int n_rowgroups, int n_tile_groups, int S,
                                                             const T* __restrict__ x, T* __restrict__ partials) {
  using V = typename VecOf<T>::type;
  constexpr int VEC = VecOf<T>::N;
  constexpr int TB = 4 / TW;
  __shared__ V lds_acc[TW > 1 ? 4 * R * WAVE : 1];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tb = wave % TB, tw = wave / TB;
  const int tg = (int)(blockIdx.x % (unsigned)n_tile_groups);
  const int64_t sg = blockIdx.x / (unsigned)n_tile_groups;
  const int64_t slot = sg * TW + tw;
  const int rg0 = (tg * TB + tb) * R;
  const int r_eff = max(0, min(R, n_rowgroups - rg0));
  const bool active = slot < S && r_eff > 0;

  V acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = (V)(T(0));

  if (active) {
    const int64_t ncb = (n + U - 1) / U;
    const T* __restrict__ a_base = A + (int64_t)rg0 * (WAVE * VEC) + lane * VEC;
    for (int64_t cb = slot; cb < ncb; cb += S) {
      const int64_t j0 = cb * U;
      T xs[U];
      int64_t jc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t j = j0 + u;
        jc[u] = j < n ? j : n - 1;
        const T xv = x[jc[u]];
        xs[u] = j < n ? xv : T(0);
      }
      V a[U][R];
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (r < r_eff) a[u][r] = nt_load(reinterpret_cast<const V*>(a_base + jc[u] * ld + r * (WAVE * VEC)));
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (r < r_eff) acc[r] += a[u][r] * xs[u];
        }
      }
    }
  }
  if constexpr (TW > 1) {
#pragma unroll
    for (int r = 0; r < R; ++r) lds_acc[(wave * R + r) * WAVE + lane] = acc[r];
    __syncthreads();
    if (tw != 0 || r_eff <= 0) return;
#pragma unroll
    for (int k = 1; k < TW; ++k) {
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] += lds_acc[((k * TB + tb) * R + r) * WAVE + lane];
    }
  } else {
    if (!active) return;
  }
  T* __restrict__ p = partials + sg * ld + (int64_t)rg0 * (WAVE * VEC) + lane * VEC;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (r < r_eff) *reinterpret_cast<V*>(p + r * (WAVE * VEC)) = acc[r];
  }
}

// Pass N for up to THREE vectors on ONE read of A: y_k = A x_k, k < 3 (pg_mat_mul_multi).  The step-size search of
// fb_tools.jl:46-55 forms A z once per halving of gamma; its candidates gamma / 2, gamma / 4, gamma / 8 differ in the n-vector z
// only, so their images can be taken together.  This is gemv_n_partial_kernel<T, 4, 2, 1> (the plan of every matrix of >= 13 row
// groups) with three x and three accumulator sets: per vector the SAME multiply-adds in the SAME order over the same slots, so each
// y_k is bit-identical to pg_mat_mul's -- the search takes the decisions it would have taken one product at a time.
template <typename T>
__global__ __launch_bounds__(256) void gemv_n3_partial_kernel(const T* __restrict__ A, int64_t ld, int64_t n, int n_rowgroups, int n_tile_groups,
                                                              int S, const T* __restrict__ x0, const T* __restrict__ x1,
                                                              const T* __restrict__ x2, T* __restrict__ partials, int64_t part_stride) {
  using V = typename VecOf<T>::type;
  constexpr int VEC = VecOf<T>::N;
  constexpr int R = 4, U = 2, TB = 4;
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tb = wave % TB;
  const int tg = (int)(blockIdx.x % (unsigned)n_tile_groups);
  const int64_t slot = blockIdx.x / (unsigned)n_tile_groups;
  const int rg0 = (tg * TB + tb) * R;
  const int r_eff = max(0, min(R, n_rowgroups - rg0));
  if (!(slot < S && r_eff > 0)) return;
  V acc[3][R];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int r = 0; r < R; ++r) acc[k][r] = (V)(T(0));
  }
  const int64_t ncb = (n + U - 1) / U;
  const T* __restrict__ a_base = A + (int64_t)rg0 * (WAVE * VEC) + lane * VEC;
  for (int64_t cb = slot; cb < ncb; cb += S) {
    const int64_t j0 = cb * U;
    T xs[3][U];
    int64_t jc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = j0 + u;
      jc[u] = j < n ? j : n - 1;
      const T v0 = x0[jc[u]], v1 = x1[jc[u]], v2 = x2[jc[u]];
      xs[0][u] = j < n ? v0 : T(0), xs[1][u] = j < n ? v1 : T(0), xs[2][u] = j < n ? v2 : T(0);
    }
    V a[U][R];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (r < r_eff) a[u][r] = nt_load(reinterpret_cast<const V*>(a_base + jc[u] * ld + r * (WAVE * VEC)));
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (r < r_eff) acc[k][r] += a[u][r] * xs[k][u];
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    T* __restrict__ p = partials + (int64_t)k * part_stride + slot * ld + (int64_t)rg0 * (WAVE * VEC) + lane * VEC;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (r < r_eff) *reinterpret_cast<V*>(p + r * (WAVE * VEC)) = acc[k][r];
    }
  }
}

// -------------------------------------------------------------------------------------------------
// pass T: g[j] = sum_{i in row chunk} A[i, j] * r[i]
// r chunk staged in LDS; each wave streams C adjacent columns, UR row groups per step.
// -------------------------------------------------------------------------------------------------
template <typename T, int C, int UR, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void gemv_t_kernel(const T* __restrict__ A, int64_t ld, int64_t n,
                                                            int64_t m, int rg_begin, int nrg,
                                                            const T* __restrict__ r, T* __restrict__ g, int line_cols) {
  using V = typename VecOf<T>::type;
  constexpr int VEC = VecOf<T>::N;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  V* lds_r = reinterpret_cast<V*>(smem_raw);

  const int64_t row0 = (int64_t)rg_begin * (WAVE * VEC);
  for (int idx = threadIdx.x; idx < nrg * WAVE; idx += WAVES * 64) {
    const int64_t i0 = row0 + (int64_t)idx * VEC;
    V v;
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = (i0 + e < m) ? r[i0 + e] : T(0);
    lds_r[idx] = v;
  }
  __syncthreads();

  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t ncg = (n + C - 1) / C;

  // The waves of a workgroup take WAVES adjacent column groups per step, and a workgroup takes those runs in chunks of
  // whole 128-byte lines of g (CgMap, pg_gemv_tn.h): dealt wave by wave round-robin, every line of g was written in
  // sixteen 8-byte pieces by sixteen waves of four workgroups -- the partial writes that cost the sweeps 3-8 % of their
  // stream (profiles/r3_mid_columns_counters.md).  PG_T_LINE_COLS = 1 restores the old deal.
  const pgtn::CgMap map((ncg + WAVES - 1) / WAVES, C * WAVES, line_cols, blockIdx.x, gridDim.x);
  for (int64_t i = 0; i < map.cnt; ++i) {
    const int64_t cg = map.at(i) * WAVES + wave;
    if (cg >= ncg) continue;  // (the last run of a matrix whose group count is no multiple of WAVES)
    const int64_t j0 = cg * C;
    const T* __restrict__ p[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int64_t j = (j0 + c < n) ? (j0 + c) : (n - 1);
      p[c] = A + j * ld + row0 + lane * VEC;
    }
    T acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = T(0);

    int rg = 0;
    for (; rg + UR <= nrg; rg += UR) {
      V a[C][UR];
#pragma unroll
      for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int u = 0; u < UR; ++u)
          a[c][u] = nt_load(reinterpret_cast<const V*>(p[c] + (int64_t)(rg + u) * (WAVE * VEC)));
      }
      V rv[UR];
#pragma unroll
      for (int u = 0; u < UR; ++u) rv[u] = lds_r[(rg + u) * WAVE + lane];
#pragma unroll
      for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int u = 0; u < UR; ++u) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[c] = fma(a[c][u][e], rv[u][e], acc[c]);
        }
      }
    }
    for (; rg < nrg; ++rg) {
      const V rv = lds_r[rg * WAVE + lane];
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const V a = nt_load(reinterpret_cast<const V*>(p[c] + (int64_t)rg * (WAVE * VEC)));
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[c] = fma(a[e], rv[e], acc[c]);
      }
    }
    // wave reduction: every lane ends with the full sums, fixed order
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = wave_allsum(acc[c]);
    T out = acc[0];
#pragma unroll
    for (int c = 1; c < C; ++c) out = (lane == c) ? acc[c] : out;
    if (lane < C && j0 + lane < n) g[j0 + lane] = out;
  }
}

// g[j] = sum_k chunks[k][j]
template <typename T>
__global__ __launch_bounds__(256) void sum_chunks_kernel(const T* __restrict__ chunks, int nchunks, int64_t n,
                                                         T* __restrict__ g) {
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (int64_t)gridDim.x * 256) {
    double a = 0.0;
    for (int k = 0; k < nchunks; ++k) a += (double)chunks[(int64_t)k * n + j];
    g[j] = (T)a;
  }
}

template <typename T>
__global__ void cast_scalar_kernel(const T* __restrict__ in, double* __restrict__ out) {
  *out = (double)(*in);
}

// r chunk per workgroup in pass T: gfx950 has 160 KiB of LDS per CU and pass T runs one workgroup per CU, so up to
// 128 KiB of r are staged at once (f32 columns of 32768 rows, f64 of 16384) before row chunking sets in.  More than
// 64 KiB of dynamic LDS has to be opted into per kernel (hipFuncSetAttribute, done once per instantiation).
constexpr int64_t LDS_R_BYTES = 128 * 1024;
constexpr int64_t LDS_DEFAULT_LIMIT = 64 * 1024;

// columns of up to this many 1 KiB row groups take the one-wave-per-column-group sweep (gemv_tnw_kernel); 0 = never.
// Set from the sweeps in profiles/r2_tune_tn_short_columns.log
constexpr int PG_TN_WAVE_MAX_RG = 8;
// ... up to this many: the workgroup-shared sweep with the lane-parallel epilogue (gemv_tnc_kernel); beyond: gemv_tn_kernel.
// Interleaved A/B, five rounds (profiles/r2_tune_tn_mid_columns.log): 17 / 20 / 24 row groups 7.03 / 6.83 / 6.96 TB/s against
// 6.27 / 6.53 / 6.88 for gemv_tn; level at 28 and 32 (the same shape re-allocated moves by 3 %: no finer cut than this)
constexpr int PG_TN_COOP_MAX_RG = 24;
// 25 .. 28: gemv_tn_kernel<4, 8, 8>; 29 .. 128: gemv_tnm_kernel (tn_mid_covers, pg_gemv_tn3.hip); beyond: teams of workgroups

// ----------------------------------------------------------------------------------------------
// host-side launch planning
// ----------------------------------------------------------------------------------------------
struct PlanN {
  int R, U, TW, n_rowgroups, n_tiles, n_tile_groups, S, S_eff;
};


// Launch geometry of pass N.  Tunables (environment, for experiments): PG_N_R, PG_N_U, PG_N_TW, PG_N_WAVES_PER_CU.
PlanN plan_n(const pg_mat* A) {
  PlanN p;
  const int64_t rows_per_rg = 1024 / (int64_t)pg_sizeof(A->dtype);
  p.n_rowgroups = (int)(A->ld / rows_per_rg);
  // Wave tile: R KiB of each of U columns per step (8 KiB of loads in flight per wave).  Measured on MI355X
  // (scripts/tune_gemv.py, profiles/r1_tune_block_counts.md): the pass peaks with ~6.5 MiB in flight chip-wide;
  // deeper queues cost 3-8 % (DRAM page conflicts / queueing), shallower ones starve the memory system.
  if (p.n_rowgroups >= 4) {
    p.R = 4;
    p.U = 2;
  } else if (p.n_rowgroups >= 2) {
    p.R = 2;
    p.U = 4;
  } else {
    p.R = 1;
    p.U = 8;
  }
  p.R = env_int("PG_N_R", p.R);
  p.U = env_int("PG_N_U", p.U);
  p.n_tiles = (p.n_rowgroups + p.R - 1) / p.R;
  // waves sharing a row tile inside a workgroup (LDS-combined): all 4 when there are fewer than 4 tiles
  p.TW = p.n_tiles >= 4 ? 1 : (p.n_tiles >= 2 ? 2 : 4);
  p.TW = env_int("PG_N_TW", p.TW);
  if (p.TW != 1 && p.TW != 2 && p.TW != 4) p.TW = 1;
  const int TB = 4 / p.TW;
  p.n_tile_groups = (p.n_tiles + TB - 1) / TB;
  const int64_t ncb = A->n > 0 ? (A->n + p.U - 1) / p.U : 1;
  // ~832 active waves chip-wide (3.25 per CU; x R*U = 8 KiB each = 6.5 MiB in flight) for tall matrices, ~896 for
  // short ones: the measured optimum of the wave-count sweeps (profiles/r1_tune_gemv.log); +-128 waves costs 2-3 %
  const int per_cu_x4 = env_int("PG_N_WAVES_PER_CU", 0) > 0 ? 4 * env_int("PG_N_WAVES_PER_CU", 0) : (p.n_tiles >= 8 ? 13 : 14);
  const int64_t target_waves = (int64_t)A->ctx->num_cu * per_cu_x4 / 4;
  const int64_t tw_abs = env_int("PG_N_WAVES", 0);
  int64_t S = (tw_abs > 0 ? tw_abs : target_waves) / ((int64_t)p.n_tile_groups * TB);
  if (S < 1) S = 1;
  if (S > ncb) S = ncb;
  S = (S + p.TW - 1) / p.TW * p.TW;
  if (S > 4096) S = 4096;
  p.S = (int)S;
  p.S_eff = (p.S + p.TW - 1) / p.TW;
  return p;
}


template <typename T, int R, int U>
pg_status launch_n_ru(pg_mat* A, const PlanN& p, const T* x) {
  const int64_t blocks64 = (int64_t)p.n_tile_groups * ((p.S + p.TW - 1) / p.TW);
  const unsigned blocks = (unsigned)blocks64;
  hipStream_t st = A->ctx->stream;
  pg_prof_scope prof(A->ctx, PG_K_GEMV_N);
  const T* Ad = (const T*)A->data;
  T* part = (T*)A->partials;
  if (p.TW == 1)
    hipLaunchKernelGGL((gemv_n_partial_kernel<T, R, U, 1>), dim3(blocks), dim3(256), 0, st, Ad, A->ld, A->n,
                       p.n_rowgroups, p.n_tile_groups, p.S, x, part);
  else if (p.TW == 2)
    hipLaunchKernelGGL((gemv_n_partial_kernel<T, R, U, 2>), dim3(blocks), dim3(256), 0, st, Ad, A->ld, A->n,
                       p.n_rowgroups, p.n_tile_groups, p.S, x, part);
  else
    hipLaunchKernelGGL((gemv_n_partial_kernel<T, R, U, 4>), dim3(blocks), dim3(256), 0, st, Ad, A->ld, A->n,
                       p.n_rowgroups, p.n_tile_groups, p.S, x, part);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

template <typename T>
pg_status launch_n_partial(pg_mat* A, const PlanN& p, const T* x) {
#define PG_N_CASE(RR, UU) \
  if (p.R == RR && p.U == UU) return launch_n_ru<T, RR, UU>(A, p, x)
  PG_N_CASE(4, 4);
  PG_N_CASE(2, 8);
  PG_N_CASE(1, 8);
  PG_N_CASE(4, 2);
  PG_N_CASE(4, 8);
  PG_N_CASE(2, 4);
  PG_N_CASE(8, 2);
  PG_N_CASE(8, 4);
  PG_N_CASE(2, 16);
  PG_N_CASE(1, 16);
  PG_N_CASE(4, 1);
  PG_N_CASE(8, 1);
  PG_N_CASE(16, 1);
  PG_N_CASE(16, 2);
  PG_N_CASE(2, 2);
#undef PG_N_CASE
  pg_set_error("no gemv_n instantiation for R=%d U=%d", p.R, p.U);
  return PG_ERR_UNSUPPORTED;
}

// y = A x - b (b nullable), y has y_len >= m valid slots (entries in [m, y_len) are zeroed);
// with_f: dscal[PG_S_F] = f_scale * ||y||^2 (and typed mirror)
template <typename T>
pg_status gemv_n(pg_mat* A, const T* x, const T* b, T* y, int64_t y_len, bool with_f, double f_scale,
                 T* f_typed) {
  pg_ctx* c = A->ctx;
  if (A->m == 0) {
    if (with_f) PG_HIP(hipMemsetAsync(c->dscal + PG_S_F, 0, sizeof(double), c->stream));
    if (with_f && f_typed) PG_HIP(hipMemsetAsync(f_typed, 0, sizeof(T), c->stream));
    return PG_OK;
  }
  PlanN p = plan_n(A);
  PG_TRY(ensure_partials(A, p.S_eff));
  if (A->n == 0) {
    PG_HIP(hipMemsetAsync(A->partials, 0, (size_t)p.S_eff * A->ld * sizeof(T), c->stream));
  } else {
    PG_TRY(launch_n_partial<T>(A, p, x));
  }
  int64_t fb = (A->ld + 63) / 64;
  if (fb > 1024) fb = 1024;  // row groups are grid-strided beyond this
  const unsigned blocks = (unsigned)fb;
  pg_prof_scope prof(c, PG_K_GEMV_N_FINISH);
  if (with_f)
    hipLaunchKernelGGL((gemv_n_finish_kernel<T, true>), dim3(blocks), dim3(1024), 0, c->stream,
                       (const T*)A->partials, A->ld, A->m, p.S_eff, b, y, y_len, f_scale, c->red_partials,
                       c->red_counter, c->dscal + PG_S_F, f_typed, ColPack<T>{});
  else
    hipLaunchKernelGGL((gemv_n_finish_kernel<T, false>), dim3(blocks), dim3(1024), 0, c->stream,
                       (const T*)A->partials, A->ld, A->m, p.S_eff, b, y, y_len, 0.0, (double*)nullptr,
                       (unsigned*)nullptr, (double*)nullptr, (T*)nullptr, ColPack<T>{});
  PG_LAUNCH_CHECK();
  return PG_OK;
}

// y_k = A x_k for nv <= 3 vectors in one read of A (see gemv_n3_partial_kernel); a missing third / second vector repeats the first
template <typename T>
pg_status gemv_n_multi(pg_mat* A, int nv, const void* const* xs, void* const* ys) {
  pg_ctx* c = A->ctx;
  PlanN p = plan_n(A);
  if (pg_row_sharded(c) || pg_col_sharded(c) || A->m <= 0 || A->n <= 0 || !(p.R == 4 && p.U == 2 && p.TW == 1)) {
    pg_set_error("the multi-vector product needs an unsharded matrix of at least 13 row groups (its pass-N plan is R=%d U=%d TW=%d)", p.R, p.U, p.TW);
    return PG_ERR_UNSUPPORTED;
  }
  PG_TRY(ensure_partials(A, 3 * p.S_eff));
  const int64_t stride = (int64_t)p.S_eff * A->ld;
  const unsigned blocks = (unsigned)((int64_t)p.n_tile_groups * p.S);
  const T* x[3] = {(const T*)xs[0], (const T*)xs[nv > 1 ? 1 : 0], (const T*)xs[nv > 2 ? 2 : 0]};
  {
    pg_prof_scope prof(c, PG_K_GEMV_N);
    hipLaunchKernelGGL((gemv_n3_partial_kernel<T>), dim3(blocks), dim3(256), 0, c->stream, (const T*)A->data, A->ld, A->n, p.n_rowgroups,
                       p.n_tile_groups, p.S, x[0], x[1], x[2], (T*)A->partials, stride);
    PG_LAUNCH_CHECK();
  }
  int64_t fb = (A->ld + 63) / 64;
  if (fb > 1024) fb = 1024;
  pg_prof_scope prof(c, PG_K_GEMV_N_FINISH);
  for (int k = 0; k < nv; ++k) {
    hipLaunchKernelGGL((gemv_n_finish_kernel<T, false>), dim3((unsigned)fb), dim3(1024), 0, c->stream, (const T*)A->partials + k * stride, A->ld,
                       A->m, p.S_eff, (const T*)nullptr, (T*)ys[k], A->m, 0.0, (double*)nullptr, (unsigned*)nullptr, (double*)nullptr,
                       (T*)nullptr, ColPack<T>{});
    PG_LAUNCH_CHECK();
  }
  return PG_OK;
}

template <typename T, int C, int UR, int WAVES>
pg_status launch_t_cuw(pg_mat* A, int rg_begin, int nrg, const T* r, T* g, int64_t col0, int64_t ncols) {
  pg_ctx* c = A->ctx;
  const size_t lds = (size_t)nrg * 1024;
  const int64_t ncg = (ncols + C - 1) / C;
  // workgroups per CU limited by LDS (160 KiB) and by 32 waves
  int lds_cap = (int)(160 * 1024 / (lds > 0 ? lds : 1));
  const int wave_cap = 32 / WAVES;
  if (lds_cap > wave_cap) lds_cap = wave_cap;
  int per_cu = env_int("PG_T_BLOCKS_PER_CU", 1);  // 1: C*UR KiB x WAVES in flight per CU is the sweet spot
  if (per_cu > lds_cap) per_cu = lds_cap;
  if (per_cu < 1) per_cu = 1;
  int64_t blocks = (int64_t)c->num_cu * per_cu;
  // long columns: 4-wave workgroups on only 3/4 of the CUs (192 x 4 waves x 16 KiB = 12 MiB in flight) measured
  // 2-3 % faster than one workgroup on every CU; all workgroups are resident, so the column groups stay balanced
  if (WAVES >= 4 && nrg >= 16 && per_cu == 1) blocks = (int64_t)c->num_cu * 3 / 4;
  if (env_int("PG_T_BLOCKS", 0) > 0) blocks = env_int("PG_T_BLOCKS", 0);
  const int64_t need = (ncg + WAVES - 1) / WAVES;
  if (blocks > need) blocks = need;
  if (blocks < 1) blocks = 1;
  if ((int64_t)lds > LDS_DEFAULT_LIMIT) {
    static std::mutex mu;  // (contexts of several host threads may launch the same instantiation for the first time)
    static bool opted_in[64] = {};
    std::lock_guard<std::mutex> lock(mu);
    const int dev = c->device & 63;
    if (!opted_in[dev]) {
      PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_t_kernel<T, C, UR, WAVES>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_R_BYTES));
      opted_in[dev] = true;
    }
  }
  pg_prof_scope prof(c, PG_K_GEMV_T);
  hipLaunchKernelGGL((gemv_t_kernel<T, C, UR, WAVES>), dim3((unsigned)blocks), dim3(WAVES * 64), lds, c->stream,
                     (const T*)A->data + col0 * A->ld, A->ld, ncols, A->m, rg_begin, nrg, r, g,
                     env_int("PG_T_LINE_COLS", 32));
  PG_LAUNCH_CHECK();
  return PG_OK;
}

// Tunables (environment, for experiments): PG_T_C, PG_T_UR, PG_T_WAVES, PG_T_BLOCKS_PER_CU.
template <typename T>
pg_status launch_t(pg_mat* A, int rg_begin, int nrg, const T* r, T* g, int64_t col0, int64_t ncols) {
  // default: 2 adjacent columns x UR row groups in flight per wave; 2-wave workgroups, one per CU, when a column
  // is long enough for UR = 8 (512 waves x 16 KiB = 8 MiB in flight chip-wide); shorter columns use more waves
  // (measured: scripts/tune_gemv.py and the block-count sweeps in profiles/r1_tune_gemv.log)
  int dC = 2, dUR = 8, dW = 4;  // columns of >= 16 row groups (4096 f32 rows): 2 columns x 8 row groups per wave
  if (nrg < 16) dW = 2;         // 8..15 row groups: the same tile, 2-wave workgroups on every CU
  if (nrg < 8) {
    dUR = 4;
    dW = 4;
  }
  if (nrg < 4) {
    dUR = 2;
    dW = 8;
  }
  if (nrg < 2) dC = 4;
  const int C = env_int("PG_T_C", dC), UR = env_int("PG_T_UR", dUR), W = env_int("PG_T_WAVES", dW);
#define PG_T_CASE(CC, UU, WW) \
  if (C == CC && UR == UU && W == WW) return launch_t_cuw<T, CC, UU, WW>(A, rg_begin, nrg, r, g, col0, ncols)
  PG_T_CASE(4, 4, 8);
  PG_T_CASE(4, 4, 4);
  PG_T_CASE(4, 4, 16);
  PG_T_CASE(2, 8, 8);
  PG_T_CASE(8, 2, 8);
  PG_T_CASE(4, 8, 8);
  PG_T_CASE(8, 4, 8);
  PG_T_CASE(2, 4, 8);
  PG_T_CASE(4, 2, 8);
  PG_T_CASE(2, 4, 16);
  PG_T_CASE(4, 2, 16);
  PG_T_CASE(2, 4, 4);
  PG_T_CASE(2, 8, 4);
  PG_T_CASE(2, 16, 4);
  PG_T_CASE(2, 16, 8);
  PG_T_CASE(1, 8, 8);
  PG_T_CASE(1, 16, 8);
  PG_T_CASE(1, 16, 4);
  PG_T_CASE(2, 2, 8);
  PG_T_CASE(2, 4, 2);
  PG_T_CASE(2, 8, 2);
  PG_T_CASE(2, 16, 2);
  PG_T_CASE(1, 16, 2);
  PG_T_CASE(2, 8, 1);
  PG_T_CASE(2, 16, 1);
  PG_T_CASE(4, 8, 2);
  PG_T_CASE(4, 4, 2);
  PG_T_CASE(4, 8, 4);
  PG_T_CASE(4, 2, 8);
#undef PG_T_CASE
  pg_set_error("no gemv_t instantiation for C=%d UR=%d WAVES=%d", C, UR, W);
  return PG_ERR_UNSUPPORTED;
}

// g = A' r ; gchunks: workspace [nchunks * n] used only when m spans several LDS chunks (may be null when
// a single chunk suffices)
template <typename T>
pg_status gemv_t(pg_mat* A, const T* r, T* g, T** gchunks_ws) {
  pg_ctx* c = A->ctx;
  if (A->n == 0) return PG_OK;
  if (A->m == 0) {
    PG_HIP(hipMemsetAsync(g, 0, (size_t)A->n * sizeof(T), c->stream));
    return PG_OK;
  }
  const int64_t rows_per_rg = 1024 / (int64_t)sizeof(T);
  const int n_rowgroups = (int)(A->ld / rows_per_rg);
  const int rg_per_chunk = (int)(LDS_R_BYTES / 1024);
  const int nchunks = (n_rowgroups + rg_per_chunk - 1) / rg_per_chunk;
  if (nchunks == 1) return launch_t<T>(A, 0, n_rowgroups, r, g, 0, A->n);
  if (*gchunks_ws == nullptr) {
    hipError_t e = hipMalloc((void**)gchunks_ws, (size_t)nchunks * A->n * sizeof(T));
    if (e != hipSuccess) {
      pg_set_error("hipMalloc for gradient chunk partials failed: %s", hipGetErrorString(e));
      return PG_ERR_ALLOC;
    }
  }
  for (int k = 0; k < nchunks; ++k) {
    const int rb = k * rg_per_chunk;
    const int nr = (rb + rg_per_chunk <= n_rowgroups) ? rg_per_chunk : (n_rowgroups - rb);
    PG_TRY(launch_t<T>(A, rb, nr, r, *gchunks_ws + (int64_t)k * A->n, 0, A->n));
  }
  int64_t blocks = (A->n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sum_chunks_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, c->stream, (const T*)*gchunks_ws,
                     nchunks, A->n, g);
  PG_LAUNCH_CHECK();
  return PG_OK;
}


// true when a single-sweep geometry covers the shape: one wave per column group (<= 8 row groups), one workgroup (<= 128)
// or a team of up to 16 workgroups (<= 1024 row groups: 262144 rows in Float32, 131072 in Float64)
template <typename T>
bool tn_supported(const pg_mat* A) {
  const int64_t rows_per_rg = 1024 / (int64_t)sizeof(T);
  const int64_t nrg = A->ld / rows_per_rg;
  return A->m > 0 && A->n > 0 && A->n < ((int64_t)1 << 31) && (tn_single_wg_supported<T>(A) || tn_team_covers((int)nrg));
}

// Geometry of the single sweep by column length.  PG_TN_KERNEL = wave | wg | team forces one (experiments, tests).
template <typename T>
pg_status launch_tn(pg_mat* A, TNArgs<T>& a, int* blocks_out) {
  const int nrg = a.nrg;
  const char* force = env_str("PG_TN_KERNEL");
  const bool single_ok = tn_single_wg_supported<T>(A);
  if (force != nullptr && *force) {
    if (force[0] == 'm')  // experiments: a named instantiation of gemv_tnm_kernel (pg_gemv_tn3.hip)
      return launch_tn_mid<T>(A, a, blocks_out, env_int("PG_TN_U", 8), env_int("PG_TN_C", 2), env_int("PG_TN_WAVES", 4),
                              env_int("PG_TN_DB", 0) + 1, env_int("PG_TN_BLOCKS_PER_CU", 1));
    if (force[0] == 'w' && force[1] == 'a' && tn_wave_covers(nrg)) return launch_tn_wave<T>(A, a, blocks_out);
    if (force[0] == 't' && tn_team_covers(nrg)) return launch_tn_team<T>(A, a, blocks_out);
    if (force[0] == 'c' && tn_coop_covers(nrg)) return launch_tn_coop<T>(A, a, blocks_out);
  } else {
    if (!single_ok) return launch_tn_team<T>(A, a, blocks_out);
    if (nrg <= PG_TN_WAVE_MAX_RG && tn_wave_covers(nrg)) return launch_tn_wave<T>(A, a, blocks_out);
    if (nrg > PG_TN_WAVE_MAX_RG && nrg <= PG_TN_COOP_MAX_RG && tn_coop_covers(nrg)) return launch_tn_coop<T>(A, a, blocks_out);
    // 29..128 row groups (the headline's 64 among them): gemv_tnm_kernel with U fitted to the column (pg_gemv_tn3.hip)
    if (single_ok && tn_mid_covers(nrg)) return launch_tn_mid_default<T>(A, a, blocks_out);
  }
  if (!single_ok) return launch_tn_team<T>(A, a, blocks_out);
  // 17..32 row groups (m = 8192 in Float32): eight waves of U = 4 with C = 8 columns per step measured 4 % faster than four
  // waves of U = 8 (775 vs 742 it/s on 8192 x 262144, profiles/r1_tune_tn_geometry.log); 33..64 stay on four waves
  // very short columns (profiles/r1_tune_tn_short_columns.log): the fewer waves share a column, the fewer cross-wave
  // exchanges per byte -- one wave per column up to 2 row groups (512 rows f32: 2.6 -> 4.6 TB/s), two waves up to 8
  // (1024 rows: 4.5 -> 5.3 TB/s, 2048 rows: 6.1 -> 6.3 TB/s); from 9 row groups on, four waves
  int W = env_int("PG_TN_WAVES", nrg <= 2 ? 1 : nrg <= 8 ? 2 : (nrg <= 16 || (nrg > 32 && nrg <= 64)) ? 4 : 8);
  if (W != 1 && W != 2 && W != 8) W = 4;
  int U = 1;
  while (U * W < nrg) U *= 2;
  // two register tiles of C * U KiB per wave (one in flight, one being consumed), one workgroup per CU: the measured
  // optimum (scripts/tune_tn.py) is C * U = 32 -- 7.0 TB/s at 16384 x 2^20, 6.8 TB/s at 8192 x 262144
  // (short columns, U <= 4: 16 KiB tiles and two workgroups per CU measured best -- 6.0 TB/s at 2048 x 2^20)
  int C = env_int("PG_TN_C", U >= 8 ? 32 / U : 16 / U);
  if (W == 8 && U == 16) C = 1;
  if (W == 8 && U == 4 && env_int("PG_TN_C", 0) == 0) C = 8;
  if (W <= 2 && env_int("PG_TN_C", 0) == 0) C = (W == 2 && U == 4) ? 4 : (U == 1 ? 16 : 8);  // (2,4,4) (2,2,8) (1,2,8) (1,1,16)
  if (sizeof(T) == 8 && U == 1 && C > 16) C = 16;  // <f64, 1, 32> would spill
#define PG_TN_CASE(UU, CC, WW) \
  if (U == UU && C == CC && W == WW) return launch_tn_ucw<T, UU, CC, WW>(A, a, blocks_out)
  PG_TN_CASE(16, 1, 4);
  PG_TN_CASE(16, 2, 4);
  PG_TN_CASE(8, 2, 4);
  PG_TN_CASE(8, 4, 4);
  PG_TN_CASE(4, 4, 4);
  PG_TN_CASE(4, 8, 4);
  PG_TN_CASE(2, 8, 4);
  PG_TN_CASE(2, 16, 4);
  PG_TN_CASE(1, 16, 4);
  PG_TN_CASE(1, 32, 4);
  PG_TN_CASE(16, 1, 8);
  PG_TN_CASE(8, 2, 8);
  PG_TN_CASE(8, 4, 8);
  PG_TN_CASE(4, 4, 8);
  PG_TN_CASE(4, 8, 8);
  // very short columns: two waves (3..8 row groups) or one wave (1..2) per column -- defaults and their tuner neighbours
  PG_TN_CASE(4, 4, 2);
  PG_TN_CASE(4, 8, 2);
  PG_TN_CASE(2, 8, 2);
  PG_TN_CASE(2, 16, 2);
  PG_TN_CASE(2, 8, 1);
  PG_TN_CASE(2, 16, 1);
  PG_TN_CASE(1, 8, 1);
  PG_TN_CASE(1, 16, 1);
  PG_TN_CASE(1, 32, 1);
  PG_TN_CASE(1, 8, 2);
  PG_TN_CASE(1, 16, 2);
#undef PG_TN_CASE
  pg_set_error("no gemv_tn instantiation for U=%d C=%d WAVES=%d", U, C, W);
  return PG_ERR_UNSUPPORTED;
}

// all-reduce helper
pg_status do_allreduce(pg_ctx* c, void* buf, int64_t count, int dtype) {
  if (!c->allreduce) {
    if (c->allreduce_begin && c->allreduce_wait) {  // only the asynchronous pair is registered
      int rc = c->allreduce_begin(c->allreduce_user, buf, count, dtype, (void*)c->stream);
      if (rc == 0) rc = c->allreduce_wait(c->allreduce_user, (void*)c->stream);
      if (rc != 0) {
        pg_set_error("all-reduce callback failed with code %d", rc);
        return PG_ERR_COLLECTIVE;
      }
    }
    return PG_OK;
  }
  int rc = c->allreduce(c->allreduce_user, buf, count, dtype, (void*)c->stream);
  if (rc != 0) {
    pg_set_error("all-reduce callback failed with code %d", rc);
    return PG_ERR_COLLECTIVE;
  }
  return PG_OK;
}

// ---- column sharding ---------------------------------------------------------------------------------------------
// This rank holds A[:, J_p] and the J_p slices of every n-vector; m-vectors are replicated.  What crosses ranks:
//   * A x: every rank's partial A[:, J_p] x[J_p] (m elements), SUM all-reduce, then - b and the norm locally;
//   * the four epilogue scalars { g(z), ||res||_inf, <g, res>, ||res||^2 }: each rank writes its values into its own
//     group of slots (four scalars, each as a hi / lo pair) of a zeroed vector; the SUM all-reduce then acts as an all-gather and every rank
//     combines the groups in rank order (sum, max, sum, sum) -- one collective, deterministic, no MAX reduction needed.
// A' r needs no collective at all (the columns are local), which is what lets the single-sweep iteration run with ONE
// all-reduce of m + 8 * nranks elements per iteration.
// Every scalar travels as a (hi, lo) pair of working-precision values, hi = (T) d, lo = (T)(d - hi): the slots of a rank
// are zero on every other rank, so the SUM passes both through exactly and d = hi + lo arrives with the ~48 (Float32) / 106
// (Float64) bits the unsharded path keeps in its fp64 scalar block -- the line search compares f(z) against the model with a
// tolerance of 10 eps, and a length n_global > 2^24 must survive the trip.
// (COL_SLOTS = 8 working-precision words per rank -- 4 scalars x (hi, lo) -- plus one shared group whose first word sums the
// ranks' team-timeout flags: ColPack in pg_gemv_tn.h)
template <typename T>
__global__ void col_pack_scalars_kernel(ColPack<T> p) {
  col_pack_slot(p, (int)(blockIdx.x * blockDim.x + threadIdx.x));
}

// slots (after the all-reduce) -> the four global scalars, combined in rank order, and the global team-timeout flag
template <typename T>
__device__ __forceinline__ void col_unpack(const T* __restrict__ slots, int nranks, double* __restrict__ s4, double* __restrict__ team_err) {
  double gz = 0.0, ri = 0.0, dg = 0.0, rs = 0.0;
  for (int p = 0; p < nranks; ++p) {
    const T* q = slots + COL_SLOTS * p;
    gz += (double)q[0] + (double)q[1];
    ri = pg_maxn(ri, (double)q[2] + (double)q[3]);
    dg += (double)q[4] + (double)q[5];
    rs += (double)q[6] + (double)q[7];
  }
  s4[0] = gz;
  s4[1] = ri;
  s4[2] = dg;
  s4[3] = rs;
  if (team_err != nullptr) {  // some rank's sweep timed out (1) or was refused at launch (2): every rank falls back in this step
    if (slots[COL_SLOTS * nranks + 1] != T(0)) *team_err = 2.0;
    else if (slots[COL_SLOTS * nranks] != T(0)) *team_err = 1.0;
  }
}

template <typename T>
__global__ void col_unpack_scalars_kernel(const T* __restrict__ slots, int nranks, double* __restrict__ s4, double* __restrict__ team_err) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  col_unpack(slots, nranks, s4, team_err);
}

// After the all-reduce of [A v partial sums ; slots]: r = payload - b over the m rows (rows m .. ld of r stay as they are:
// zero), f = f_scale ||r||^2 -> f_dst (fixed summation order), and -- the same launch -- the scalar slots combined into
// dscal[PG_S_GZ ..].  One kernel where there were two (residual combination, unpack).
template <typename T>
__global__ __launch_bounds__(256) void col_combine_kernel(const T* __restrict__ payload, const T* __restrict__ b, T* __restrict__ r,
                                                          int64_t m, double f_scale, double* __restrict__ red_partials,
                                                          unsigned* __restrict__ red_counter, double* __restrict__ f_dst,
                                                          const T* __restrict__ slots, int nranks, double* __restrict__ s4,
                                                          double* __restrict__ team_err) {
  double acc[1] = {0.0};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
    const T o = payload[i] - b[i];
    r[i] = o;
    acc[0] += (double)o * (double)o;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) col_unpack(slots, nranks, s4, team_err);
  const double ps[1] = {f_scale};
  grid_reduce_finalize<1, 0u, 4>(acc, red_partials, red_counter, f_dst, ps);
}

template <typename T>
pg_status col_ensure_cbuf(pg_ls* f) {
  if (f->cbuf) return PG_OK;
  const size_t bytes = ((size_t)f->A->ld + COL_SLOTS * ((size_t)f->ctx->shard_nranks + 1) + 64) * sizeof(T);
  hipError_t e = hipMalloc(&f->cbuf, bytes);
  if (e != hipSuccess) {
    pg_set_error("hipMalloc for the column-sharding payload failed: %s", hipGetErrorString(e));
    return PG_ERR_ALLOC;
  }
  PG_HIP(hipMemsetAsync(f->cbuf, 0, bytes, f->ctx->stream));  // no slot is ever read before it was written, whatever the rank count
  return PG_OK;
}

// dscal[PG_S_GZ..PG_S_RESSQ]: local values in, global values out
template <typename T>
pg_status col_allreduce_scalars_t(pg_ls* f) {
  pg_ctx* c = f->ctx;
  PG_TRY(col_ensure_cbuf<T>(f));
  T* slots = (T*)f->cbuf + f->A->ld;
  const int nr = c->shard_nranks;
  const int nslots = COL_SLOTS * (nr + 1);
  ColPack<T> pk;
  pk.slots = slots;
  pk.nranks = nr;
  pk.rank = c->shard_rank;
  pk.s4 = c->dscal + PG_S_GZ;
  pk.team_err = nullptr;  // no sweep in flight on this path
  hipLaunchKernelGGL(col_pack_scalars_kernel<T>, dim3((nslots + 63) / 64), dim3(64), 0, c->stream, pk);
  PG_LAUNCH_CHECK();
  PG_TRY(do_allreduce(c, slots, nslots, f->A->dtype));
  hipLaunchKernelGGL(col_unpack_scalars_kernel<T>, dim3(1), dim3(64), 0, c->stream, (const T*)slots, nr, c->dscal + PG_S_GZ,
                     (double*)nullptr);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

template <typename T>
pg_status ls_residual_t(pg_ls* f, const T* x) {
  pg_mat* A = f->A;
  if (pg_col_sharded(f->ctx)) {  // r = sum_p A[:, J_p] x[J_p] - b
    PG_TRY(gemv_n<T>(A, x, (const T*)nullptr, (T*)f->r, A->ld, false, 0.0, (T*)nullptr));
    PG_TRY(do_allreduce(f->ctx, f->r, A->m, A->dtype));
    PG_TRY(pg_residual_combo_async(f->ctx, A->dtype, A->m, f->r, 1.0, f->r, -1.0, f->b, 0.5 * f->lam, nullptr));
    f->a_passes += 1;
    f->r_gen++;
    return PG_OK;
  }
  T* f_typed = pg_row_sharded(f->ctx) ? ((T*)f->gbuf + A->n) : nullptr;
  PG_TRY(gemv_n<T>(A, x, (const T*)f->b, (T*)f->r, A->ld, true, 0.5 * f->lam, f_typed));
  f->a_passes += 1;
  f->r_gen++;
  return PG_OK;
}

template <typename T>
pg_status ls_value_t(pg_ls* f, const T* x) {
  PG_TRY(ls_residual_t<T>(f, x));
  pg_ctx* c = f->ctx;
  if (pg_row_sharded(c)) {
    T* ft = (T*)f->gbuf + f->A->n;
    PG_TRY(do_allreduce(c, ft, 1, f->A->dtype));
    hipLaunchKernelGGL(cast_scalar_kernel<T>, dim3(1), dim3(1), 0, c->stream, (const T*)ft, c->dscal + PG_S_F);
    PG_LAUNCH_CHECK();
  }
  return PG_OK;
}

// One sweep: g = lam A' r (r = f->r, the residual of x), epilogue for (x, g, gamma), v = z + beta (z - z_old),
// then f->r = A v - b and dscal[PG_S_F] = lam/2 ||A v - b||^2; epilogue scalars -> dscal[PG_S_GZ..PG_S_RESSQ]
template <typename T>
pg_status ls_fused_pass_t(pg_ls* f, const T* r_src, T* r_dst, double* f_dst, const T* x, const T* z_old, double gamma,
                          double beta, int g_kind, double g_p0, double g_p1, T* g_out, T* y, T* z_new, T* res, T* v_out,
                          const T* g_v0, const T* g_v1) {
  pg_ctx* c = f->ctx;
  pg_mat* A = f->A;
  // row shards: only as a row TEAM (pg_ctx_set_row_team): the devices exchange the per-column partial dots inside the sweep
  const bool rteam = pg_row_sharded(c) && pg_rteam_active(c);
  if (rteam) {
    PG_TRY(pg_mat_row_team_agree(c, A));  // (a collective the first time: every device of the team makes this call)
    if (!tn_peer_covers(A->team_nrg)) {
      pg_set_error("a row-team sweep covers row blocks of at most %d rows per device (the team's longest has %d row groups)",
                   (int)(64 * (1024 / sizeof(T))), A->team_nrg);
      return PG_ERR_UNSUPPORTED;
    }
  }
  if ((pg_row_sharded(c) && !rteam) || !tn_supported<T>(A)) {
    pg_set_error("the single-sweep pass needs an unsharded or column-sharded operator (or a row team) with at most %d rows",
                 (int)(1024 * (1024 / sizeof(T))));  // 1024 row groups: teams of up to 16 workgroups (pg_gemv_tn2.hip)
    return PG_ERR_UNSUPPORTED;
  }
  const bool cols = pg_col_sharded(c);
  if (cols) PG_TRY(col_ensure_cbuf<T>(f));
  TNArgs<T> a;
  a.A = (const T*)A->data;
  a.ld = A->ld;
  a.n = A->n;
  a.m = A->m;
  a.nrg = (int)(A->ld / (1024 / (int64_t)sizeof(T)));
  a.r = r_src;
  a.x = x;
  a.z_old = z_old;
  const T gm = (T)gamma;
  a.gamma = gm;
  a.beta = (T)beta;
  a.p0 = g_kind == PG_G_NORML1 ? (T)(gm * (T)g_p0) : (T)g_p0;
  a.p1 = (T)g_p1;
  a.lam_ls = (T)f->lam;
  a.g_kind = g_kind;
  a.p0v = g_kind == PG_G_INDBOX || g_kind == PG_G_NORML1 ? g_v0 : nullptr;  // IndBox: lo_j; NormL1: weights lam_j
  a.p1v = g_kind == PG_G_INDBOX ? g_v1 : nullptr;
  a.gscale = g_kind == PG_G_NORML1 ? (a.p0v != nullptr ? 1.0 : (double)(T)g_p0) : 0.0;
  a.g_out = g_out;
  a.y = y;
  a.z_new = z_new;
  a.res = res;
  a.v_out = v_out;
  a.partials = nullptr;
  a.red_partials = c->red_partials;
  a.red_counter = c->red_counter;
  a.scal_out = c->dscal + PG_S_GZ;
  a.line_cols = env_int("PG_TN_LINE_COLS", 0) > 0 ? env_int("PG_TN_LINE_COLS", 0) : 32;  // experiments: 1 = dealt one by one (rounds 1-2)
  int blocks = 0;
  const pg_status launched = rteam ? launch_tn_peer<T>(A, a, &blocks) : launch_tn<T>(A, a, &blocks);
  // Column shards: a sweep refused on THIS rank only (a cooperative launch that does not fit next to something else on the
  // device) must not leave the peers alone in this step's all-reduce.  The rank posts the same collective with an empty
  // m-vector and its refused flag set; every rank then sees PG_ERR_UNSUPPORTED at the scalar read-back, drops the step
  // and leaves the single-sweep mode together (pg_iter.hip::redo_with_two_sweeps).
  const bool refused = launched == PG_ERR_UNSUPPORTED && cols;
  if (refused) blocks = 0;
  else PG_TRY(launched);
  if (!refused) f->a_passes += 1;
  int64_t fb = (A->ld + 63) / 64;
  if (fb > 1024) fb = 1024;
  if (cols) {
    // [sum of the workgroup partials of A[:, J_p] v[J_p] ; this rank's four scalars in its slots ; team-timeout flag] ->
    // ONE all-reduce -> r = . - b and its norm ; scalars combined in rank order.  Three launches beside the sweep:
    // finish (+ pack), the collective, combine (+ unpack).
    T* payload = (T*)f->cbuf;
    const int nr = c->shard_nranks;
    ColPack<T> pk;
    pk.slots = payload + A->ld;
    pk.nranks = nr;
    pk.rank = c->shard_rank;
    pk.s4 = c->dscal + PG_S_GZ;
    pk.team_err = c->dscal + PG_S_TEAMERR;
    pk.refused = refused ? 1 : 0;
    {
      pg_prof_scope prof(c, PG_K_GEMV_N_FINISH);
      hipLaunchKernelGGL((gemv_n_finish_kernel<T, false>), dim3((unsigned)fb), dim3(1024), 0, c->stream,
                         (const T*)A->partials, A->ld, A->m, blocks, (const T*)nullptr, payload, A->ld, 0.0,
                         (double*)nullptr, (unsigned*)nullptr, (double*)nullptr, (T*)nullptr, pk);
      PG_LAUNCH_CHECK();
    }
    PG_TRY(do_allreduce(c, payload, A->ld + COL_SLOTS * (nr + 1), A->dtype));
    int64_t cb = (A->m + 1023) / 1024;
    if (cb > 256) cb = 256;
    if (cb < 1) cb = 1;
    hipLaunchKernelGGL(col_combine_kernel<T>, dim3((unsigned)cb), dim3(256), 0, c->stream, (const T*)payload, (const T*)f->b, r_dst,
                       A->m, 0.5 * f->lam, c->red_partials, c->red_counter, f_dst, (const T*)(payload + A->ld), nr,
                       c->dscal + PG_S_GZ, c->dscal + PG_S_TEAMERR);
    PG_LAUNCH_CHECK();
    if (r_dst == (T*)f->r) f->r_gen++;
    return PG_OK;
  }
  {
    pg_prof_scope prof(c, PG_K_GEMV_N_FINISH);
    hipLaunchKernelGGL((gemv_n_finish_kernel<T, true>), dim3((unsigned)fb), dim3(1024), 0, c->stream,
                       (const T*)A->partials, A->ld, A->m, blocks, (const T*)f->b, r_dst,
                       r_dst == (T*)f->r ? A->ld : A->m /* only f->r is padded to ld */, 0.5 * f->lam,
                       c->red_partials, c->red_counter, rteam ? c->rteam.f_local : f_dst, (T*)nullptr, ColPack<T>{});
    PG_LAUNCH_CHECK();
  }
  // row team: f = sum_p 1/2 lam ||A_p v - b_p||^2 and the devices' timeout flags, exchanged like the dots (no collective)
  if (rteam) PG_TRY(peer_scalar_exchange(c, c->rteam.f_local, f_dst != nullptr ? f_dst : c->rteam.f_local));
  if (r_dst == (T*)f->r) f->r_gen++;
  return PG_OK;
}

// Matrix-level form of the single sweep for x -> f(A x) compositions (PANOC, panoc.jl:184,197-199 and the A z of the
// next line search :43): g = A' r for a caller-supplied m-vector r (= grad f(A x)), the epilogue for (x, g, gamma), and
// Az = A z_new (no b, no norm) from the columns while they are in registers.
template <typename T>
pg_status mat_fused_tn_t(pg_mat* A, const T* r, const T* x, double gamma, int g_kind, double g_p0, double g_p1, T* g_out,
                         T* y, T* z_new, T* res, T* Az_out, bool image_of_res) {
  pg_ctx* c = A->ctx;
  if (pg_row_sharded(c) || pg_col_sharded(c) || !tn_supported<T>(A)) {
    pg_set_error("the single-sweep pass needs an unsharded operator with at most %d rows", (int)(1024 * (1024 / sizeof(T))));
    return PG_ERR_UNSUPPORTED;
  }
  if (A->rpad == nullptr) {  // r zero-padded to the leading dimension (the kernel reads whole 1 KiB row groups)
    PG_HIP(hipMalloc(&A->rpad, (size_t)A->ld * sizeof(T)));
    PG_HIP(hipMemsetAsync(A->rpad, 0, (size_t)A->ld * sizeof(T), c->stream));
  }
  PG_HIP(hipMemcpyAsync(A->rpad, r, (size_t)A->m * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
  TNArgs<T> a;
  a.A = (const T*)A->data;
  a.ld = A->ld;
  a.n = A->n;
  a.m = A->m;
  a.nrg = (int)(A->ld / (1024 / (int64_t)sizeof(T)));
  a.r = (const T*)A->rpad;
  a.x = x;
  a.z_old = x;
  const T gm = (T)gamma;
  a.gamma = gm;
  a.beta = T(0);
  a.v_is_res = image_of_res ? 1 : 0;  // v = res = x - z  |  v = z
  a.p0 = g_kind == PG_G_NORML1 ? (T)(gm * (T)g_p0) : (T)g_p0;
  a.p1 = (T)g_p1;
  a.lam_ls = T(1);
  a.g_kind = g_kind;
  a.gscale = g_kind == PG_G_NORML1 ? (double)(T)g_p0 : 0.0;
  a.g_out = g_out;
  a.y = y;
  a.z_new = z_new;
  a.res = res;
  a.v_out = nullptr;
  a.partials = nullptr;
  a.red_partials = c->red_partials;
  a.red_counter = c->red_counter;
  a.scal_out = c->dscal + PG_S_GZ;
  a.line_cols = env_int("PG_TN_LINE_COLS", 0) > 0 ? env_int("PG_TN_LINE_COLS", 0) : 32;  // experiments: 1 = dealt one by one (rounds 1-2)
  int blocks = 0;
  PG_TRY(launch_tn<T>(A, a, &blocks));
  int64_t fb = (A->ld + 63) / 64;
  if (fb > 1024) fb = 1024;
  pg_prof_scope prof(c, PG_K_GEMV_N_FINISH);
  hipLaunchKernelGGL((gemv_n_finish_kernel<T, false>), dim3((unsigned)fb), dim3(1024), 0, c->stream,
                     (const T*)A->partials, A->ld, A->m, blocks, (const T*)nullptr, Az_out, A->m, 0.0, (double*)nullptr,
                     (unsigned*)nullptr, (double*)nullptr, (T*)nullptr, ColPack<T>{});
  PG_LAUNCH_CHECK();
  return PG_OK;
}

// TWO instances of mat_fused_tn_t on ONE read of A (gemv_tnm_pair_kernel, pg_gemv_tn3.hip): the same gamma and g, two pairs
// (r, x); every output twice; scalars -> dscal[PG_S_PAIR .. + 8).  ZeroFPR's line search (zerofpr.jl:200-217) evaluates its
// trial points x = xbar_prev + tau d one sweep each; this carries tau and tau / 2 through the same pass over A.
template <typename T>
pg_status mat_fused_tn_pair_t(pg_mat* A, const T* r1, const T* x1, const T* r2, const T* x2, double gamma, int g_kind, double g_p0, double g_p1,
                              T* g1, T* y1, T* z1, T* res1, T* Az1, T* g2, T* y2, T* z2, T* res2, T* Az2, bool image_of_res) {
  pg_ctx* c = A->ctx;
  const int nrg = (int)(A->ld / (1024 / (int64_t)sizeof(T)));
  if (pg_row_sharded(c) || pg_col_sharded(c) || !tn_supported<T>(A) || !tn_pair_covers(nrg)) {
    pg_set_error("the two-point sweep needs an unsharded operator with %d .. %d rows", (int)(32 * (1024 / sizeof(T)) + 1), (int)(64 * (1024 / sizeof(T))));
    return PG_ERR_UNSUPPORTED;
  }
  for (void** pad : {&A->rpad, &A->rpad2}) {
    if (*pad == nullptr) {
      PG_HIP(hipMalloc(pad, (size_t)A->ld * sizeof(T)));
      PG_HIP(hipMemsetAsync(*pad, 0, (size_t)A->ld * sizeof(T), c->stream));
    }
  }
  PG_HIP(hipMemcpyAsync(A->rpad, r1, (size_t)A->m * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
  PG_HIP(hipMemcpyAsync(A->rpad2, r2, (size_t)A->m * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
  TNArgs<T> a;
  a.A = (const T*)A->data;
  a.ld = A->ld;
  a.n = A->n;
  a.m = A->m;
  a.nrg = nrg;
  a.r = (const T*)A->rpad;
  a.x = x1;
  a.z_old = x1;
  const T gm = (T)gamma;
  a.gamma = gm;
  a.beta = T(0);
  a.v_is_res = image_of_res ? 1 : 0;  // both instances: Az = A z  |  A (x - z)
  a.p0 = g_kind == PG_G_NORML1 ? (T)(gm * (T)g_p0) : (T)g_p0;
  a.p1 = (T)g_p1;
  a.lam_ls = T(1);
  a.g_kind = g_kind;
  a.gscale = g_kind == PG_G_NORML1 ? (double)(T)g_p0 : 0.0;
  a.g_out = g1;
  a.y = y1;
  a.z_new = z1;
  a.res = res1;
  a.v_out = nullptr;
  a.partials = nullptr;
  a.red_partials = c->red_partials;
  a.red_counter = c->red_counter;
  a.scal_out = c->dscal + PG_S_PAIR;
  a.line_cols = 32;
  int blocks = 0;
  T* partials2 = nullptr;
  PG_TRY(launch_tn_pair<T>(A, a, (const T*)A->rpad2, x2, g2, y2, z2, res2, &blocks, &partials2));
  int64_t fb = (A->ld + 63) / 64;
  if (fb > 1024) fb = 1024;
  pg_prof_scope prof(c, PG_K_GEMV_N_FINISH);
  for (int k = 0; k < 2; ++k) {
    hipLaunchKernelGGL((gemv_n_finish_kernel<T, false>), dim3((unsigned)fb), dim3(1024), 0, c->stream,
                       k == 0 ? (const T*)A->partials : (const T*)partials2, A->ld, A->m, blocks, (const T*)nullptr, k == 0 ? Az1 : Az2, A->m,
                       0.0, (double*)nullptr, (unsigned*)nullptr, (double*)nullptr, (T*)nullptr, ColPack<T>{});
    PG_LAUNCH_CHECK();
  }
  return PG_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void scale_kernel(T* __restrict__ v, int64_t n, T a) {
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (int64_t)gridDim.x * 256) v[j] *= a;
}

template <typename T>
pg_status ls_grad_stage_t(pg_ls* f, T* grad_out) {
  // second half of a gradient evaluation: g = lam A' r from the residual already in f->r (and f already in
  // dscal[PG_S_F] / gbuf[n]), then the all-reduce of [grad ; f] when the rows are sharded
  pg_ctx* c = f->ctx;
  pg_mat* A = f->A;
  const bool sharded = pg_row_sharded(c);  // column shards own their columns: A' r is local
  T* gdst = sharded ? (T*)f->gbuf : grad_out;
  const int64_t rows_per_rg = 1024 / (int64_t)sizeof(T);
  const int n_rowgroups = (int)(A->ld / rows_per_rg);
  // Pipelined collective: pass T runs in K column chunks; the all-reduce of chunk k is issued
  // asynchronously (RCCL's own stream) as soon as its columns are done and overlaps pass T of chunk k+1; only the
  // last chunk's collective is exposed.  The payload's trailing f rides with the last chunk.
  int K = env_int("PG_ALLREDUCE_CHUNKS", 4);
  const bool pipelined = sharded && c->allreduce_begin != nullptr && c->allreduce_wait != nullptr && K > 1 && A->m > 0 &&
                         (int64_t)n_rowgroups * 1024 <= LDS_R_BYTES && A->n >= (int64_t)K * 4096;
  if (pipelined) {
    const int64_t per = (A->n / K + 255) / 256 * 256;  // keep chunk starts 1 KiB aligned
    for (int k = 0; k < K; ++k) {
      const int64_t c0 = (int64_t)k * per;
      if (c0 >= A->n) break;
      const bool is_last = (k == K - 1) || (c0 + per >= A->n);
      const int64_t nc = is_last ? (A->n - c0) : per;
      PG_TRY(launch_t<T>(A, 0, n_rowgroups, (const T*)f->r, gdst + c0, c0, nc));
      if (f->lam != 1.0) {
        int64_t blocks = (nc + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(scale_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, c->stream, gdst + c0, nc, (T)f->lam);
        PG_LAUNCH_CHECK();
      }
      int rc = c->allreduce_begin(c->allreduce_user, gdst + c0, nc + (is_last ? 1 : 0), A->dtype, (void*)c->stream);
      if (rc != 0) {
        pg_set_error("asynchronous all-reduce callback failed with code %d", rc);
        return PG_ERR_COLLECTIVE;
      }
      if (is_last) break;
    }
    f->a_passes += 1;
    int rc = c->allreduce_wait(c->allreduce_user, (void*)c->stream);
    if (rc != 0) {
      pg_set_error("all-reduce wait callback failed with code %d", rc);
      return PG_ERR_COLLECTIVE;
    }
  } else {
    T* chunks = (T*)f->gchunks;
    PG_TRY(gemv_t<T>(A, (const T*)f->r, gdst, &chunks));
    f->gchunks = chunks;
    f->a_passes += 1;
    if (f->lam != 1.0 && A->n > 0) {
      int64_t blocks = (A->n + 255) / 256;
      if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL(scale_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, c->stream, gdst, A->n, (T)f->lam);
      PG_LAUNCH_CHECK();
    }
    if (sharded) PG_TRY(do_allreduce(c, f->gbuf, A->n + 1, A->dtype));
  }
  if (sharded) {
    hipLaunchKernelGGL(cast_scalar_kernel<T>, dim3(1), dim3(1), 0, c->stream, (const T*)f->gbuf + A->n,
                       c->dscal + PG_S_F);
    PG_LAUNCH_CHECK();
    if (A->n > 0)
      PG_HIP(hipMemcpyAsync(grad_out, f->gbuf, (size_t)A->n * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
  }
  return PG_OK;
}

template <typename T>
pg_status ls_vg_t(pg_ls* f, const T* x, T* grad_out) {
  PG_TRY(ls_residual_t<T>(f, x));
  return ls_grad_stage_t<T>(f, grad_out);
}

}  // namespace

pg_status pg_ls_fused_pass_async(pg_ls* f, const void* r_src, void* r_dst, double* f_dst, const void* x, const void* z_old,
                                 double gamma, double beta, int g_kind, double g_p0, double g_p1, void* grad, void* y,
                                 void* z_new, void* res, void* v_next, const void* g_v0, const void* g_v1) {
  if (r_src == nullptr) r_src = f->r;
  if (r_dst == nullptr) r_dst = f->r;
  if (f_dst == nullptr) f_dst = f->ctx->dscal + PG_S_F;
  return f->A->dtype == PG_F32
             ? ls_fused_pass_t<float>(f, (const float*)r_src, (float*)r_dst, f_dst, (const float*)x, (const float*)z_old, gamma,
                                      beta, g_kind, g_p0, g_p1, (float*)grad, (float*)y, (float*)z_new, (float*)res,
                                      (float*)v_next, (const float*)g_v0, (const float*)g_v1)
             : ls_fused_pass_t<double>(f, (const double*)r_src, (double*)r_dst, f_dst, (const double*)x, (const double*)z_old,
                                       gamma, beta, g_kind, g_p0, g_p1, (double*)grad, (double*)y, (double*)z_new,
                                       (double*)res, (double*)v_next, (const double*)g_v0, (const double*)g_v1);
}

pg_status pg_ls_allreduce_epilogue_scalars(pg_ls* f) {
  if (!pg_col_sharded(f->ctx)) return PG_OK;
  return f->A->dtype == PG_F32 ? col_allreduce_scalars_t<float>(f) : col_allreduce_scalars_t<double>(f);
}

// What every device of a row team must know identically before its first sweep over a matrix: the LONGEST row block of the team
// in row groups.  It sizes the sweep (same U, same workgroups per compute unit, same column map on every device), and it decides
// whether the team sweeps at all: a block beyond the sweep's 64 row groups on ANY device and none does.  (Round 4 decided that
// per device from its own block: 131073 rows over eight devices are one block of 65 row groups and seven of 64 -- one device
// would have run two sweeps and all-reduced n + 1 elements while its peers polled inboxes it never filled.)  Agreed ONCE per
// matrix and team through the REGISTERED all-reduce -- a sum over one-hot slots, i.e. an all-gather, so no MAX reduction is
// asked of the host's collective -- when the first iterator over the matrix is created (or at a first bare sweep): every device
// makes the same calls in the same order, so the collective pairs up, and unlike an exchange through the inboxes it cannot come
// back differently on different devices.
pg_status pg_mat_row_team_agree(pg_ctx* c, pg_mat* A) {
  if (A->team_nrg != 0 && A->team_nrg_gen == c->rteam.gen) return PG_OK;
  PG_REQUIRE(pg_rteam_active(c) && c->rteam.f_local != nullptr && pg_row_sharded(c), "the context is not a row team with a registered all-reduce");
  const int64_t nrg = A->ld / (1024 / (int64_t)pg_sizeof(A->dtype));
  double hd[16] = {};
  float hf[16] = {};
  hd[c->rteam.rank] = (double)nrg, hf[c->rteam.rank] = (float)nrg;  // (row groups of one device: far below 2^24)
  void* slots = (void*)(c->rteam.f_local + 2);
  const size_t bytes = 16 * pg_sizeof(A->dtype);
  PG_HIP(hipStreamSynchronize(c->stream));
  PG_HIP(hipMemcpy(slots, A->dtype == PG_F64 ? (const void*)hd : (const void*)hf, bytes, hipMemcpyHostToDevice));
  PG_TRY(do_allreduce(c, slots, 16, A->dtype));
  PG_HIP(hipStreamSynchronize(c->stream));
  PG_HIP(hipMemcpy(A->dtype == PG_F64 ? (void*)hd : (void*)hf, slots, bytes, hipMemcpyDeviceToHost));
  double longest = 0.0;
  for (int q = 0; q < c->rteam.n; ++q) longest = fmax(longest, A->dtype == PG_F64 ? hd[q] : (double)hf[q]);
  A->team_nrg = (int)longest;
  A->team_nrg_gen = c->rteam.gen;
  return PG_OK;
}

bool pg_ls_fused_pass_supported(const pg_ls* f) {
  pg_ctx* c = f->ctx;
  if (pg_row_sharded(c)) {  // as a row team only (pg_gemv_tn4.hip), and only when EVERY device's block is covered
    // PURE: reads what pg_mat_row_team_agree left on the matrix.  The agreement is a collective with a status of its own --
    // pg_iter_create and the bare sweep make it an explicit step and propagate its error (round 5 ran it from inside this
    // predicate and mapped a failure to "unsupported": one rank on two sweeps, its peers polling inboxes nobody fills)
    if (!(pg_rteam_active(c) && f->A->m > 0 && f->A->n > 0)) return false;
    if (!(f->A->team_nrg != 0 && f->A->team_nrg_gen == c->rteam.gen)) return false;
    return tn_peer_covers(f->A->team_nrg);
  }
  return f->A->dtype == PG_F32 ? tn_supported<float>(f->A) : tn_supported<double>(f->A);
}

pg_status pg_ls_grad_stage_async(pg_ls* f, void* grad_out) {
  return f->A->dtype == PG_F32 ? ls_grad_stage_t<float>(f, (float*)grad_out) : ls_grad_stage_t<double>(f, (double*)grad_out);
}
pg_status pg_ls_residual_async(pg_ls* f, const void* x) {
  return f->A->dtype == PG_F32 ? ls_residual_t<float>(f, (const float*)x) : ls_residual_t<double>(f, (const double*)x);
}
pg_status pg_ls_value_async(pg_ls* f, const void* x) {
  return f->A->dtype == PG_F32 ? ls_value_t<float>(f, (const float*)x) : ls_value_t<double>(f, (const double*)x);
}
pg_status pg_ls_vg_async(pg_ls* f, const void* x, void* grad_out) {
  return f->A->dtype == PG_F32 ? ls_vg_t<float>(f, (const float*)x, (float*)grad_out)
                               : ls_vg_t<double>(f, (const double*)x, (double*)grad_out);
}

extern "C" {

pg_status pg_mat_mul(pg_mat* A, const void* x, void* y) {
  PG_REQUIRE(A != nullptr, "matrix is null");
  PG_REQUIRE((A->n == 0 || x != nullptr) && (A->m == 0 || y != nullptr), "null vector");
  if (A->dtype == PG_F32)
    return gemv_n<float>(A, (const float*)x, nullptr, (float*)y, A->m, false, 0.0, nullptr);
  return gemv_n<double>(A, (const double*)x, nullptr, (double*)y, A->m, false, 0.0, nullptr);
}

pg_status pg_mat_mul_multi(pg_mat* A, int32_t nv, const void* const* xs, void* const* ys) {
  PG_REQUIRE(A != nullptr && xs != nullptr && ys != nullptr, "null argument");
  PG_REQUIRE(nv >= 1 && nv <= 3, "one to three vectors");
  for (int k = 0; k < nv; ++k) PG_REQUIRE(xs[k] != nullptr && ys[k] != nullptr, "null vector");
  return A->dtype == PG_F32 ? gemv_n_multi<float>(A, nv, xs, ys) : gemv_n_multi<double>(A, nv, xs, ys);
}

pg_status pg_mat_mul_adjoint(pg_mat* A, const void* r, void* g) {
  PG_REQUIRE(A != nullptr, "matrix is null");
  PG_REQUIRE((A->m == 0 || r != nullptr) && (A->n == 0 || g != nullptr), "null vector");
  // chunk workspace is cached on a throw-away pointer here (rare path: m > 16384 rows f32)
  if (A->dtype == PG_F32) {
    float* ws = nullptr;
    pg_status s = gemv_t<float>(A, (const float*)r, (float*)g, &ws);
    if (ws) {
      (void)hipStreamSynchronize(A->ctx->stream);
      (void)hipFree(ws);
    }
    return s;
  }
  double* ws = nullptr;
  pg_status s = gemv_t<double>(A, (const double*)r, (double*)g, &ws);
  if (ws) {
    (void)hipStreamSynchronize(A->ctx->stream);
    (void)hipFree(ws);
  }
  return s;
}

pg_status pg_ls_create(pg_ctx* c, pg_mat* A, const void* b, double lam, pg_ls** out) {
  PG_REQUIRE(c != nullptr && A != nullptr && out != nullptr, "null argument");
  PG_REQUIRE(A->ctx == c, "matrix belongs to another context");
  PG_REQUIRE(A->m == 0 || b != nullptr, "b is null");
  *out = nullptr;
  pg_ls* f = new pg_ls();
  f->ctx = c;
  f->A = A;
  f->b = b;
  f->lam = lam;
  const size_t es = pg_sizeof(A->dtype);
  if (hipMalloc(&f->r, (size_t)A->ld * es) != hipSuccess ||
      hipMalloc(&f->gbuf, (size_t)(A->n + 1) * es) != hipSuccess) {
    pg_set_error("LeastSquares workspace allocation failed");
    pg_ls_destroy(f);
    return PG_ERR_ALLOC;
  }
  PG_HIP(hipMemsetAsync(f->r, 0, (size_t)A->ld * es, c->stream));
  PG_HIP(hipMemsetAsync(f->gbuf, 0, (size_t)(A->n + 1) * es, c->stream));
  *out = f;
  return PG_OK;
}

pg_status pg_ls_destroy(pg_ls* f) {
  if (!f) return PG_OK;
  if (f->r) (void)hipFree(f->r);
  if (f->gbuf) (void)hipFree(f->gbuf);
  if (f->gchunks) (void)hipFree(f->gchunks);
  if (f->cbuf) (void)hipFree(f->cbuf);
  delete f;
  return PG_OK;
}

static pg_status mat_fused_tn_any(pg_mat* A, const void* r, const void* x, double gamma, int32_t g_kind, double g_p0, double g_p1,
                                  void* At_r, void* y, void* z, void* res, void* Av, double* scalars_out, bool image_of_res) {
  PG_REQUIRE(A != nullptr, "matrix is null");
  PG_REQUIRE(r && x && At_r && y && z && res && Av, "null vector");
  PG_REQUIRE(g_kind == PG_G_ZERO || g_kind == PG_G_NORML1 || g_kind == PG_G_INDBOX, "unknown g_kind");
  PG_REQUIRE(gamma > 0, "gamma must be positive");
  PG_TRY(A->dtype == PG_F32 ? mat_fused_tn_t<float>(A, (const float*)r, (const float*)x, gamma, g_kind, g_p0, g_p1,
                                                    (float*)At_r, (float*)y, (float*)z, (float*)res, (float*)Av, image_of_res)
                            : mat_fused_tn_t<double>(A, (const double*)r, (const double*)x, gamma, g_kind, g_p0, g_p1,
                                                     (double*)At_r, (double*)y, (double*)z, (double*)res, (double*)Av,
                                                     image_of_res));
  if (scalars_out) {
    PG_TRY(pg_read_scalars(A->ctx, PG_S_GZ, 4));
    for (int k = 0; k < 4; ++k) scalars_out[k] = A->ctx->hscal[PG_S_GZ + k];
  }
  return PG_OK;
}

pg_status pg_mat_fused_tn(pg_mat* A, const void* r, const void* x, double gamma, int32_t g_kind, double g_p0, double g_p1,
                          void* At_r, void* y, void* z, void* res, void* Az, double* scalars_out) {
  return mat_fused_tn_any(A, r, x, gamma, g_kind, g_p0, g_p1, At_r, y, z, res, Az, scalars_out, false);
}

pg_status pg_mat_fused_tn_res(pg_mat* A, const void* r, const void* x, double gamma, int32_t g_kind, double g_p0, double g_p1,
                              void* At_r, void* y, void* z, void* res, void* Ares, double* scalars_out) {
  return mat_fused_tn_any(A, r, x, gamma, g_kind, g_p0, g_p1, At_r, y, z, res, Ares, scalars_out, true);
}

static pg_status mat_fused_tn_pair_any(pg_mat* A, const void* r1, const void* x1, const void* r2, const void* x2, double gamma, int32_t g_kind,
                                       double g_p0, double g_p1, void* At_r1, void* y1, void* z1, void* res1, void* Az1, void* At_r2, void* y2,
                                       void* z2, void* res2, void* Az2, double* scalars_out, bool image_of_res) {
  PG_REQUIRE(A != nullptr, "matrix is null");
  PG_REQUIRE(r1 && x1 && r2 && x2 && At_r1 && y1 && z1 && res1 && Az1 && At_r2 && y2 && z2 && res2 && Az2, "null vector");
  PG_REQUIRE(g_kind == PG_G_ZERO || g_kind == PG_G_NORML1 || g_kind == PG_G_INDBOX, "unknown g_kind");
  PG_REQUIRE(gamma > 0, "gamma must be positive");
  PG_TRY(A->dtype == PG_F32
             ? mat_fused_tn_pair_t<float>(A, (const float*)r1, (const float*)x1, (const float*)r2, (const float*)x2, gamma, g_kind, g_p0, g_p1,
                                          (float*)At_r1, (float*)y1, (float*)z1, (float*)res1, (float*)Az1, (float*)At_r2, (float*)y2,
                                          (float*)z2, (float*)res2, (float*)Az2, image_of_res)
             : mat_fused_tn_pair_t<double>(A, (const double*)r1, (const double*)x1, (const double*)r2, (const double*)x2, gamma, g_kind, g_p0,
                                           g_p1, (double*)At_r1, (double*)y1, (double*)z1, (double*)res1, (double*)Az1, (double*)At_r2,
                                           (double*)y2, (double*)z2, (double*)res2, (double*)Az2, image_of_res));
  if (scalars_out) {
    PG_TRY(pg_read_scalars(A->ctx, PG_S_PAIR, 8));
    for (int k = 0; k < 8; ++k) scalars_out[k] = A->ctx->hscal[PG_S_PAIR + k];
  }
  return PG_OK;
}

pg_status pg_mat_fused_tn_pair(pg_mat* A, const void* r1, const void* x1, const void* r2, const void* x2, double gamma, int32_t g_kind,
                               double g_p0, double g_p1, void* At_r1, void* y1, void* z1, void* res1, void* Az1, void* At_r2, void* y2,
                               void* z2, void* res2, void* Az2, double* scalars_out) {
  return mat_fused_tn_pair_any(A, r1, x1, r2, x2, gamma, g_kind, g_p0, g_p1, At_r1, y1, z1, res1, Az1, At_r2, y2, z2, res2, Az2, scalars_out, false);
}

pg_status pg_mat_fused_tn_pair_res(pg_mat* A, const void* r1, const void* x1, const void* r2, const void* x2, double gamma, int32_t g_kind,
                                   double g_p0, double g_p1, void* At_r1, void* y1, void* z1, void* res1, void* Ares1, void* At_r2, void* y2,
                                   void* z2, void* res2, void* Ares2, double* scalars_out) {
  return mat_fused_tn_pair_any(A, r1, x1, r2, x2, gamma, g_kind, g_p0, g_p1, At_r1, y1, z1, res1, Ares1, At_r2, y2, z2, res2, Ares2, scalars_out, true);
}

pg_status pg_ls_fused_pass(pg_ls* f, const void* x, const void* z_old, double gamma, double beta, int32_t g_kind,
                           double g_p0, double g_p1, void* grad, void* y, void* z_new, void* res, void* v_next,
                           double* scalars_out) {
  PG_REQUIRE(f != nullptr, "operator is null");
  PG_REQUIRE(x && z_old && grad && y && z_new && res && v_next, "null vector");
  PG_REQUIRE(g_kind == PG_G_ZERO || g_kind == PG_G_NORML1 || g_kind == PG_G_INDBOX, "unknown g_kind");
  PG_REQUIRE(gamma > 0, "gamma must be positive");
  PG_TRY(pg_ls_fused_pass_async(f, nullptr, nullptr, nullptr, x, z_old, gamma, beta, g_kind, g_p0, g_p1, grad, y, z_new, res,
                                v_next));
  if (scalars_out) {
    PG_TRY(pg_read_scalars(f->ctx, PG_S_F, 5));
    for (int k = 0; k < 5; ++k) scalars_out[k] = f->ctx->hscal[PG_S_F + k];
  }
  return PG_OK;
}

pg_status pg_ls_value_and_gradient(pg_ls* f, const void* x, void* grad_out, double* f_out) {
  PG_REQUIRE(f != nullptr, "f is null");
  PG_REQUIRE(f->A->n == 0 || (x != nullptr && grad_out != nullptr), "null vector");
  PG_TRY(pg_ls_vg_async(f, x, grad_out));
  if (f_out) {
    PG_TRY(pg_read_scalars(f->ctx, PG_S_F, 1));
    *f_out = f->ctx->hscal[PG_S_F];
  }
  return PG_OK;
}

pg_status pg_ls_gradient(pg_ls* f, void* grad_out, const void* x, double* f_out) {
  return pg_ls_value_and_gradient(f, x, grad_out, f_out);
}

pg_status pg_ls_value(pg_ls* f, const void* x, double* f_out) {
  PG_REQUIRE(f != nullptr, "f is null");
  PG_REQUIRE(f->A->n == 0 || x != nullptr, "null vector");
  PG_TRY(pg_ls_value_async(f, x));
  if (f_out) {
    PG_TRY(pg_read_scalars(f->ctx, PG_S_F, 1));
    *f_out = f->ctx->hscal[PG_S_F];
  }
  return PG_OK;
}

pg_status pg_ls_residual_ptr(pg_ls* f, const void** r_out) {
  PG_REQUIRE(f != nullptr && r_out != nullptr, "null argument");
  *r_out = f->r;
  return PG_OK;
}

}  // extern "C"
