// The single-sweep kernel (pass TN) and its launcher, shared by the translation units that instantiate it:
// pg_gemv.hip (MODE 0: the forward-backward epilogue -- the headline kernel) and pg_gemv_dys.hip (MODE 1: the
// Davis-Yin epilogue), so that the headline instantiations are compiled on their own.
#pragma once
#include <cstdlib>

#include "pg_cgmap.h"
#include "pg_internal.h"

namespace pgtn {

constexpr int WAVE = 64;

template <typename V>
__device__ __forceinline__ V nt_load(const V* p) {
  return __builtin_nontemporal_load(p);
}

#include "pg_lanes.h"

// Wave-wide sum in every lane, no LDS round trip.  The four intra-row steps are DPP moves: quad_perm xor 1 and
// xor 2, then row_half_mirror / row_mirror (lane i <-> 7-i / 15-i: after the quad steps every lane of a quad holds
// the quad sum, so the mirrored partner contributes exactly the other quad / the other half-row); the two
// cross-row steps (xor 16, xor 32) are v_permlane16_swap / v_permlane32_swap exchanges (pg_lanes.h) -- the same pairs
// and, addition being commutative, the same bits as the ds_bpermute shuffles they replace (rounds 1-5: two LDS round
// trips per column and step, ~200 cycles nobody hides where a compute unit runs one wave per SIMD).  Fixed order =>
// deterministic.
template <typename T>
__device__ __forceinline__ T wave_allsum(T v) {
  v += pg_dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]  (xor 1)
  v += pg_dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]  (xor 2)
  v += pg_dpp_mov<0x141>(v);  // row_half_mirror
  v += pg_dpp_mov<0x140>(v);  // row_mirror
  v = swap16_add(v, v);       // rows 0 <-> 1, 2 <-> 3  (xor 16)
  v = swap32_add(v, v);       // lanes l <-> l + 32     (xor 32)
  return v;
}

// -------------------------------------------------------------------------------------------------
// pass TN: ONE sweep over A for BOTH orientations of a proximal-gradient iteration (unsharded operator).
// For a column j everything the next residual needs from it is known as soon as g_j = A_j' r is:
//     g_j -> y_j = x_j - gamma g_j -> z_j = prox(y_j) -> res_j = x_j - z_j -> v_j = z_j + beta (z_j - zold_j)
// (forward_backward.jl:117-120 / fast_forward_backward.jl:140-142 followed by :135 of the NEXT iteration), so
// A_j v_j is accumulated while the column is still in registers and A is read once per iteration instead of twice.
// A workgroup's WAVES waves share every column: wave w owns the 1 KiB row groups w * U + u (u < U) -- its slice
// of r and of the next residual live in registers for the whole kernel (no LDS staging) -- and the C column dot
// products of a step meet in LDS (one workgroup barrier per step, fixed summation order).  Each workgroup leaves
// a partial of A v in partials[blockIdx.x], reduced by gemv_n_finish_kernel like pass N's slots; thread c of the
// workgroup writes column c's outputs and accumulates the epilogue scalars for the grid reduction.
// -------------------------------------------------------------------------------------------------
// row group owned by (wave, u): every wave streams a CONTIGUOUS run of U KiB of each column (measured +1.3 % over
// interleaving the waves' row groups: longer bursts per wave)
#define TN_RG(u, wave, U, WAVES) ((wave) * (U) + (u))

template <typename T>
struct TNArgs {
  const T* A;
  int64_t ld, n, m;
  int nrg;  // 1 KiB row groups of a column
  int ueff = 0;  // team kernel: row groups per wave = ceil(nrg / (members * waves)) <= U, the rows dealt evenly over the team
  int deal_even = 0;  // team kernel: waves hold floor or ceil of nrg / (members * waves) row groups (pg_gemv_tnt.h) instead of ueff each
  const T* r;      // [ld] A x - b (lam not applied)
  const T* x;      // [n]
  const T* z_old;  // [n] the prox output of the previous iteration (for v); may alias nothing written here
  T gamma, beta, p0, p1, lam_ls;  // p0 = gamma * lam (NormL1) | lo (IndBox) ; p1 = hi
  // the vector whose image the sweep accumulates: v_j = z_j + beta (z_j - zold_j), the (extrapolated) next point
  // (fast_forward_backward.jl:135) -- or, v_is_res != 0, v_j = res_j = x_j - z_j, the forward-backward residual itself
  // (PANOC's image slab wants A res as a PRODUCT of the small vector, not as a difference of two large images)
  int v_is_res = 0;
  int g_kind;
  // NormL1 with PER-ELEMENT weights (ProximalOperators.NormL1(lambda::AbstractArray)): lam_j = p0v[j] (p1v unused, gscale = 1);
  // IndBox with PER-ELEMENT bounds (ProximalOperators.IndBox(lo::AbstractArray, hi::AbstractArray)): lo_j = p0v[j], hi_j = p1v[j];
  // nullptr: the scalars p0 / p1.  Two more n-vector streams next to x and z_old (< 0.1 % of the sweep's bytes).
  const T* p0v = nullptr;
  const T* p1v = nullptr;
  double gscale;  // lam for NormL1 (1 with per-element weights) else 0
  T *g_out, *y, *z_new, *res, *v_out;  // [n] each
  T* partials;                         // [gridDim.x][ld]
  double* red_partials;
  unsigned* red_counter;
  double* scal_out;  // 4 doubles: g(z), ||res||_inf, <g, res>, ||res||^2
  // Davis-Yin mode (MODE = 1): x = xg (the prox_g point the gradient was taken at), z_old = the splitting variable z;
  // per column  z_half = 2 xg - z - gamma g ; xh = prox_{gamma h}(z_half) ; res = xh - xg ; z+ = z + relax res ;
  // xg+ = prox_{gamma g}(z+)  -- outputs y = z_half, xh_out = xh, res, v_out = z+, z_new = xg+ ; A xg+ is accumulated
  int h_kind = 0;
  T h_p0 = T(0), h_p1 = T(0), relax = T(1);  // h_p0 = gamma * lam (NormL1) | lo (IndBox) | 1 / (1 + lam gamma) (SqrNormL2)
  T* xh_out = nullptr;
  // workgroup teams (gemv_tnt_kernel, columns longer than one workgroup's registers): team_size workgroups split the rows
  // of every column; their per-column partial dots meet in xch, a ring of tagged 8-byte granules per team
  int team_size = 1, nteams = 0;
  int line_cols = 32;  // columns a workgroup (wave, team) takes in a row before it jumps: see CgMap
  unsigned tag_base = 0;  // launch epoch << 24: the tag of step i is tag_base + i + 1, so granules of earlier launches never match
  unsigned long long* xch = nullptr;  // [nteams][ring][team_size * C * granules-per-value]
  double* team_err = nullptr;         // set to 1 when a team member gave up waiting (bounded spin)
  // PEER teams (row blocks on several devices, pg_gemv_tn4.hip): this device's index, the number of devices, and every
  // device's inbox ring as mapped into THIS device's address space (peer_ring[peer_rank] == xch)
  int peer_n = 0, peer_rank = 0;
  unsigned long long* peer_ring[16] = {};
  unsigned long long* wait_stats = nullptr;  // PEER: { waves that found their granules late, polls they spent waiting, DELAY: ticks of slack left, steps counted } (telemetry)
  long long spin_limit = 1 << 21;  // polls before a waiting wave gives up (~ seconds): bounded, never a hang (row teams: pg_ctx_row_team_tune "SPIN")
  unsigned delay_ticks = 0;  // PEER, DELAY instantiations (tests): a step's granules are accepted once every member's stamp is this old (100 MHz ticks)
#ifdef PG_TNT_EXPERIMENT
  int dbg = 0;  // timing experiments of the team kernel (wrong results): see pg_gemv_tn2.hip
#endif
};

// in-kernel prox kinds: PG_G_ZERO / PG_G_NORML1 / PG_G_INDBOX and, for the second operator of the Davis-Yin mode,
// PG_G_SQRNORML2 (p0 = the scaling 1 / (1 + lam gamma))
template <typename T>
__device__ __forceinline__ T tn_prox(int kind, T y, T p0, T p1) {
  if (kind == PG_G_NORML1) return y <= -p0 ? y + p0 : (y >= p0 ? y - p0 : T(0));
  if (kind == PG_G_INDBOX) return fmin(p1, fmax(p0, y));
  if (kind == PG_G_SQRNORML2) return y * p0;
  return y;
}

template <typename T, int U, int C, int WAVES>
struct TNTile {
  using V = typename VecOf<T>::type;
  static constexpr int VEC = VecOf<T>::N;
  V col[C][U];

  // this wave's row groups wave * U + u of the C columns of group cg
  __device__ __forceinline__ void load(const TNArgs<T>& a, int64_t cg, int wave, int lane) {
    const int64_t j0 = cg * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int64_t j = (j0 + c < a.n) ? (j0 + c) : (a.n - 1);
      const T* __restrict__ p = a.A + j * a.ld + lane * VEC;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int rg = TN_RG(u, wave, U, WAVES);
        if (rg < a.nrg) {
          col[c][u] = nt_load(reinterpret_cast<const V*>(p + (int64_t)rg * (WAVE * VEC)));
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) col[c][u][e] = T(0);
        }
      }
    }
  }
};

template <typename T, int U, int C, int WAVES, bool DOUBLE_BUFFER, int MODE = 0>
__global__ __launch_bounds__(WAVES * 64) void gemv_tn_kernel(TNArgs<T> a) {
  using V = typename VecOf<T>::type;
  constexpr int VEC = VecOf<T>::N;
  __shared__ T sm_dot[2][C][WAVES];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t ncg = (a.n + C - 1) / C;

  // this wave's rows: row groups rg(u) = wave * U + u
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

  // one column group: dot products -> workgroup totals (fixed wave order) -> epilogue -> next-residual accumulation
  auto process = [&](const TNTile<T, U, C, WAVES>& t, int64_t cg, int buf) {
    const int64_t j0 = cg * C;
    // the per-column scalars are fetched before the barrier so that their latency hides behind the dot products
    T xs[C], zos[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int64_t jc = (j0 + c < a.n) ? (j0 + c) : (a.n - 1);
      xs[c] = a.x[jc];
      zos[c] = a.z_old[jc];
    }
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
      const T xj = xs[c], zo = zos[c];
      T vj;
      if constexpr (MODE == 1) {  // Davis-Yin: davis_yin.jl:74-80 for column j, then the next prox_g point
        const T zh = T(2) * xj - zo - a.gamma * g;
        const T xh = tn_prox<T>(a.h_kind, zh, a.h_p0, a.h_p1);
        const T rj = xh - xj;
        const T zs = zo + a.relax * rj;
        const T xg = tn_prox<T>(a.g_kind, zs, a.p0, a.p1);
        vj = valid ? xg : T(0);
        if ((int)threadIdx.x == c && valid) {
          a.g_out[j] = g;
          a.y[j] = zh;
          a.xh_out[j] = xh;
          a.res[j] = rj;
          a.v_out[j] = zs;
          a.z_new[j] = xg;
          acc[1] = pg_maxn(acc[1], fabs((double)rj));
          acc[2] += (double)g * (double)rj;
          acc[3] += (double)rj * (double)rj;
        }
      } else {
        const T yj = xj - a.gamma * g;
        T zj;
        if (a.g_kind == PG_G_NORML1) {
          T th = a.p0;
          if (a.p0v != nullptr) th = pg_l1w_threshold(a.gamma, a.p0v[valid ? j : a.n - 1]);  // per-element weights lam_j
          zj = yj <= -th ? yj + th : (yj >= th ? yj - th : T(0));
        } else if (a.g_kind == PG_G_INDBOX) {
          T lo = a.p0, hi = a.p1;
          if (a.p0v != nullptr) lo = a.p0v[valid ? j : a.n - 1], hi = a.p1v[valid ? j : a.n - 1];
          zj = fmin(hi, fmax(lo, yj));
        } else
          zj = yj;
        const T rj = xj - zj;
        vj = valid ? (a.v_is_res ? rj : zj + a.beta * (zj - zo)) : T(0);
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
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) racc[u][e] = fma(t.col[c][u][e], vj, racc[u][e]);
      }
    }
    // Pin the accumulators here: their only use is at the end of the kernel, and the compiler otherwise sinks this
    // group's multiply-adds into the next group's code, keeping this tile alive across it (found on the team kernel,
    // pg_gemv_tn2.hip: whole tiles spilled; here: the <16,2,4> instantiation filled all 512 registers)
#pragma unroll
    for (int u = 0; u < U; ++u) asm volatile("" : "+v"(racc[u]));
  };

  // two register tiles: the loads of the next column group are in flight while the current one is reduced, exchanged
  // through LDS and folded into the next residual (the kernel would otherwise idle the memory system at every barrier)
  // column groups go to workgroups in chunks of whole output lines, the chunks strided across the grid: all workgroups
  // sweep one moving window of A (CgMap).  (A fully blocked assignment -- every workgroup streaming its own contiguous
  // region -- measured 3 % faster on a freshly booted device and 3-10 % slower, alternating from process to process, on
  // others: 256 far-apart streams depend on how the 64 GiB allocation happens to be mapped.)
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
  } else {  // 8-wave workgroups have half the registers per wave: one tile, two workgroups per CU overlap instead
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

// -------------------------------------------------------------------------------------------------
// pass N, stage 2: y[i] = sum_s partials[s][i] - b[i]   (i < m; fixed summation order, fp64), optional
// f = f_scale * sum_i y[i]^2 -> f_out (+ typed copy for the all-reduce payload).
// 1024 threads = 64 rows x 16 slot groups; row groups are grid-strided.
// -------------------------------------------------------------------------------------------------
// Column shards: the scalar slots behind the m-vector of the all-reduce payload (pg_gemv.hip, "column sharding").  Every rank
// owns COL_SLOTS working-precision words -- its four epilogue scalars as (hi, lo) pairs -- that are zero on all other ranks,
// so the SUM all-reduce acts as an all-gather; one more group of COL_SLOTS words behind them is shared: word 0 carries this
// rank's team-timeout flag (PG_S_TEAMERR), word 1 its launch-refused flag; their sums tell EVERY rank that some rank's sweep
// failed (they must fall back together: the two-sweep retry issues a different collective).
constexpr int COL_SLOTS = 8;
template <typename T>
struct ColPack {
  T* slots = nullptr;  // [COL_SLOTS * nranks + COL_SLOTS]; nullptr: nothing to pack
  int nranks = 0, rank = 0;
  const double* s4 = nullptr;        // this rank's { g(z), ||res||_inf, <g, res>, ||res||^2 }
  const double* team_err = nullptr;  // this rank's PG_S_TEAMERR
  int refused = 0;                   // this rank's sweep was refused at launch (nothing ran): word 1 of the shared group
};
template <typename T>
__device__ __forceinline__ void col_pack_slot(const ColPack<T>& p, int t) {
  if (t >= COL_SLOTS * (p.nranks + 1)) return;
  T v = T(0);
  if (t / COL_SLOTS == p.rank) {
    const double d = p.s4[(t % COL_SLOTS) >> 1];
    const T hi = (T)d;
    v = (t & 1) ? (T)(d - (double)hi) : hi;
  } else if (t == COL_SLOTS * p.nranks) {
    v = (p.team_err != nullptr && *p.team_err != 0.0) ? T(1) : T(0);
  } else if (t == COL_SLOTS * p.nranks + 1) {
    v = p.refused ? T(1) : T(0);
  }
  p.slots[t] = v;
}

template <typename T, bool WITH_F>
__global__ __launch_bounds__(1024) void gemv_n_finish_kernel(const T* __restrict__ partials, int64_t ld, int64_t m,
                                                             int S, const T* __restrict__ b, T* __restrict__ y,
                                                             int64_t y_len, double f_scale,
                                                             double* __restrict__ red_partials,
                                                             unsigned* __restrict__ red_counter,
                                                             double* __restrict__ f_out, T* __restrict__ f_out_typed,
                                                             ColPack<T> pack) {
  __shared__ double sm_rows[16][64];
  const int rx = threadIdx.x & 63;
  const int sg = threadIdx.x >> 6;
  double sq = 0.0;
  // column shards: the scalar slots of the all-reduce payload ride in this launch (they were their own kernel before)
  if (pack.slots != nullptr && blockIdx.x == 0)
    for (int t = (int)threadIdx.x; t < COL_SLOTS * (pack.nranks + 1); t += (int)blockDim.x) col_pack_slot(pack, t);
  for (int64_t row0 = (int64_t)blockIdx.x * 64; row0 < ld; row0 += (int64_t)gridDim.x * 64) {
    const int64_t i = row0 + rx;
    double acc = 0.0;
    if (i < ld) {
      // eight loads in flight, added in slab order (the same sum as a plain loop: 1024 slabs of a 512-row problem took 20 us
      // as a chain of dependent load -> add steps)
      int s = sg;
      for (; s + 7 * 16 < S; s += 8 * 16) {
        T v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = partials[(int64_t)(s + 16 * k) * ld + i];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += (double)v[k];
      }
      for (; s < S; s += 16) acc += (double)partials[(int64_t)s * ld + i];
    }
    sm_rows[sg][rx] = acc;
    __syncthreads();
    if (sg == 0) {
      double v = sm_rows[0][rx];
#pragma unroll
      for (int k = 1; k < 16; ++k) v += sm_rows[k][rx];
      T out = T(0);
      if (i < m) {
        if (b != nullptr) v -= (double)b[i];
        out = (T)v;
        sq += (double)out * (double)out;
      }
      if (i < y_len) y[i] = out;
    }
    __syncthreads();
  }
  if constexpr (WITH_F) {
    // wave 0 holds the row contributions; all 16 waves take part in the grid reduction
    double v[1] = {sg == 0 ? sq : 0.0};
    const double ps[1] = {f_scale};
    double fin[1];
    const bool last = grid_reduce_finalize<1, 0u, 16>(v, red_partials, red_counter, f_out, ps, fin);
    // mirror f in working precision (slot n of the all-reduce payload)
    if (last && threadIdx.x == 0 && f_out_typed != nullptr) *f_out_typed = (T)fin[0];
  }
}

using ::pg_tuning_enabled;
using ::env_str;
inline int env_int(const char* name, int dflt) {
  if (!pg_tuning_enabled()) return dflt;
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

inline pg_status ensure_partials(pg_mat* A, int S) {
  if (A->partials && A->partials_slots >= S) return PG_OK;
  if (A->partials) {
    PG_HIP(hipStreamSynchronize(A->ctx->stream));
    PG_HIP(hipFree(A->partials));
    A->partials = nullptr;
  }
  const size_t bytes = (size_t)S * (size_t)A->ld * pg_sizeof(A->dtype);
  hipError_t e = hipMalloc(&A->partials, bytes);
  if (e != hipSuccess) {
    pg_set_error("hipMalloc(%zu) for GEMV partial sums failed: %s", bytes, hipGetErrorString(e));
    return PG_ERR_ALLOC;
  }
  A->partials_slots = S;
  return PG_OK;
}

// ---- pass TN launcher ------------------------------------------------------------------------------------------
// Tunables (environment, for experiments): PG_TN_C (columns per step), PG_TN_BLOCKS_PER_CU, PG_TN_BLOCKS.
template <typename T, int U, int C, int WAVES, int MODE = 0>
pg_status launch_tn_ucw(pg_mat* A, TNArgs<T>& a, int* blocks_out) {
  pg_ctx* c = A->ctx;
  const int64_t ncg = (A->n + C - 1) / C;
  // workgroups per CU (measured, scripts/tune_tn.py): one for the big double-buffered tiles, two for 4-wave short
  // columns, four / eight for the 2- and 1-wave workgroups of very short columns
  const int bpc_default = WAVES == 1 ? 8 : WAVES == 2 ? 4 : (((WAVES == 4 && U >= 8) || (WAVES == 8 && U == 4)) ? 1 : 2);
  int64_t blocks = (int64_t)c->num_cu * env_int("PG_TN_BLOCKS_PER_CU", bpc_default);
  if (env_int("PG_TN_BLOCKS", 0) > 0) blocks = env_int("PG_TN_BLOCKS", 0);
  if (blocks > ncg) blocks = ncg;
  if (blocks > PG_RED_MAX_BLOCKS) blocks = PG_RED_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  PG_TRY(ensure_partials(A, (int)blocks));
  a.partials = (T*)A->partials;
  *blocks_out = (int)blocks;
  pg_prof_scope prof(c, PG_K_GEMV_TN);
  hipLaunchKernelGGL((gemv_tn_kernel<T, U, C, WAVES, (WAVES <= 4), MODE>), dim3((unsigned)blocks), dim3(WAVES * 64), 0,
                     c->stream, a);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

// true when the shape is covered by one workgroup per column group: every wave keeps U <= 16 row groups of r and of the
// next residual in registers
template <typename T>
bool tn_single_wg_supported(const pg_mat* A) {
  const int64_t rows_per_rg = 1024 / (int64_t)sizeof(T);
  const int64_t nrg = A->ld / rows_per_rg;
  return A->m > 0 && A->n > 0 && nrg <= 16 * 8;
}

// the other two geometries of the sweep (pg_gemv_tn2.hip): one wave per column group for short columns, teams of
// workgroups for columns longer than one workgroup's registers
bool tn_wave_covers(int nrg);
bool tn_team_covers(int nrg);
bool tn_coop_covers(int nrg);
template <typename T>
pg_status launch_tn_wave(pg_mat* A, TNArgs<T>& a, int* blocks_out);
template <typename T>
pg_status launch_tn_team(pg_mat* A, TNArgs<T>& a, int* blocks_out);
template <typename T>
pg_status launch_tn_coop(pg_mat* A, TNArgs<T>& a, int* blocks_out);
// PEER teams: the same sweep with the rows of a column on several DEVICES (pg_gemv_tn4.hip; pg_ctx_set_row_team)
bool tn_peer_covers(int nrg);
template <typename T>
pg_status launch_tn_peer(pg_mat* A, TNArgs<T>& a, int* blocks_out);
pg_status peer_scalar_exchange(pg_ctx* c, const double* f_local, double* f_out);
// gemv_tnm_kernel: 29 .. 128 row groups, the headline's 64 among them (pg_gemv_tn3.hip)
bool tn_mid_covers(int nrg);
bool tn_pair_covers(int nrg);
// two instances of the sweep on one read of A (gemv_tnm_pair_kernel, pg_gemv_tn3.hip): the second instance's inputs / outputs
template <typename T>
pg_status launch_tn_pair(pg_mat* A, TNArgs<T>& a, const T* r2, const T* x2, T* g2, T* y2, T* z2, T* res2, int* blocks_out, T** partials2_out);
template <typename T>
pg_status launch_tn_mid(pg_mat* A, TNArgs<T>& a, int* blocks_out, int U, int C, int W, int nt, int bpc);
template <typename T>
pg_status launch_tn_mid_default(pg_mat* A, TNArgs<T>& a, int* blocks_out);

}  // namespace pgtn
