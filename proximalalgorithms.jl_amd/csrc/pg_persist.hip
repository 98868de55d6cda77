// Persistent solver kernels: the FB / FFB driver loop of src/ProximalAlgorithms.jl:114-123 inside ONE
// kernel launch -- pg_iter_run_small (one workgroup) and pg_iter_run_coop (cooperating workgroups, grid barriers).
// The control flow is the one of iter_step in pg_iter.hip (forward_backward.jl:86-123, fast_forward_backward.jl:106-145,
// fb_tools.jl:24-63, nesterov.jl), restated once in solver_loop<T, Ops> and instantiated for the two back ends.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "pg_iter_internal.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Persistent solvers for launch-bound sizes: the whole driver loop
// (src/ProximalAlgorithms.jl:114-123) -- stop rule, line search, Nesterov recurrences, both GEMV orientations, prox --
// runs inside ONE kernel launch.  Scalars are computed redundantly by every thread from reduced values (so control
// flow is uniform without broadcasts).  Same control flow as iter_step above.  Two back ends share the loop:
//   * SmallOps: one 1024-thread workgroup; vectors are exchanged between phases through global memory + workgroup
//     barriers (m * n <= 2^20).
//   * CoopOps: W workgroups (one per CU, cooperative launch) that meet at grid barriers; A stays L2-resident (each
//     workgroup keeps reading the same slices), the three residual vectors live in every workgroup's LDS, and a
//     fixed-step iteration costs two grid barriers (pass N partials | combine + A' r + prox + scalar partials).
// ---------------------------------------------------------------------------------------------
// -DPG_COOP_TRACE: thread 0 of workgroup 0 stamps wall_clock64() at marked points of ONE iteration into the result
// block (diagnostics only; the marks compile to nothing otherwise)
#ifdef PG_COOP_TRACE
#define PG_MARK(ops, id) (ops).mark(id)
#else
#define PG_MARK(ops, id) ((void)0)
#endif
constexpr int SMALL_OUT_DOUBLES = 96;

template <typename T>
struct SmallParams {
  const T* A;
  long long ld;
  int m, n;
  const T* b;
  T* buf[8];  // roles at entry: 0 x, 1 grad_f_x, 2 y, 3 z, 4 res, 5 z_prev | grad_f_z, 6 rz, 7 rz_prev (6,7 optional)
  T* r;       // residual scratch (m)
  int fast, adaptive, reuse, g_kind, seq_kind, has_fixed_gamma;
  T g_p0, g_p1, lam_ls;
  T gamma, f_x, g_z, res_inf, dot_gr, res_sq, fixed_gamma;
  T min_gamma, reduce_gamma, increase_gamma, mf, seq_p0, seq_p1;
  SeqState<T> seq;
  long long k_start, maxit;
  T tol;
  double* out;  // [32] results, mapped host memory
  // cooperative back end only
  int W, nrb, m_pad;        // workgroups; 64-row blocks; padded m
  int cols_per, ncol_pad;   // columns owned by a workgroup (padded: LDS slice size)
  int two_stage, rows_per;  // row-sliced combination of the pass-N partials (large W * m)
  T* rfull;                 // [2][m_pad] the combined residual of the two-stage path
  double* npart;            // [2][W][m_pad] pass-N partial sums (double buffered by pass parity)
  double* spart;            // [2][W][4] scalar partials (double buffered by reduction parity)
  unsigned long long* bar;  // arrival counter of the grid barrier (zeroed by the host before the launch)
  int* abort_flag;
};

constexpr int SMALL_THREADS = 1024;
constexpr int SMALL_WAVES = SMALL_THREADS / 64;

// all-thread block reduction of 4 doubles (bit k of MAXMASK: max); every thread returns with the results.
// DPP / readlane wave reductions, the 16 wave totals meet in LDS and are reduced again inside a 16-lane row.
template <unsigned MAXMASK>
__device__ __forceinline__ void small_block_reduce(double (&v)[4], double* sm_red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v[0] = pg_wave_allreduce<(MAXMASK & 1u) != 0>(v[0]);
  v[1] = pg_wave_allreduce<(MAXMASK & 2u) != 0>(v[1]);
  v[2] = pg_wave_allreduce<(MAXMASK & 4u) != 0>(v[2]);
  v[3] = pg_wave_allreduce<(MAXMASK & 8u) != 0>(v[3]);
  __syncthreads();  // sm_red free
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) sm_red[wave * 4 + k] = v[k];
  }
  __syncthreads();
  static_assert(SMALL_WAVES == 16, "the second stage reduces one wave total per lane of a 16-lane row");
  v[0] = pg_row_allreduce<(MAXMASK & 1u) != 0>(sm_red[(lane & 15) * 4 + 0]);
  v[1] = pg_row_allreduce<(MAXMASK & 2u) != 0>(sm_red[(lane & 15) * 4 + 1]);
  v[2] = pg_row_allreduce<(MAXMASK & 4u) != 0>(sm_red[(lane & 15) * 4 + 2]);
  v[3] = pg_row_allreduce<(MAXMASK & 8u) != 0>(sm_red[(lane & 15) * 4 + 3]);
}

template <typename T>
__device__ __forceinline__ T small_soft(T x, T gl) {
  return x <= -gl ? x + gl : (x >= gl ? x - gl : T(0));
}

// y = x - gamma g ; z = prox(y) ; res = x - z for one element; accumulates { |z|, max|res|, g res, res^2 }
template <typename T>
__device__ __forceinline__ void small_epilogue_elem(const SmallParams<T>& p, T xv, T gv, T gamma, T gl, T& yv, T& zv,
                                                    T& rv, double (&acc)[4]) {
  yv = xv - gamma * gv;
  if (p.g_kind == PG_G_NORML1)
    zv = small_soft(yv, gl);
  else if (p.g_kind == PG_G_INDBOX)
    zv = fmin(p.g_p1, fmax(p.g_p0, yv));
  else
    zv = yv;
  rv = xv - zv;
  if (p.g_kind == PG_G_NORML1) acc[0] += fabs((double)zv);
  acc[1] = pg_maxn(acc[1], fabs((double)rv));
  acc[2] += (double)gv * (double)rv;
  acc[3] += (double)rv * (double)rv;
}

template <typename T>
__device__ __forceinline__ void small_epilogue_out(const SmallParams<T>& p, const double (&acc)[4], double (&out)[4]) {
  out[0] = p.g_kind == PG_G_NORML1 ? acc[0] * (double)p.g_p0 : 0.0;
  out[1] = acc[1];
  out[2] = acc[2];
  out[3] = acc[3];
}

// nesterov.jl recurrences evaluated by ONE wave and broadcast through LDS: their fp64 divisions and square roots are
// slow when all 16 waves of the workgroup queue for the same SIMDs with identical work
template <typename T>
__device__ __forceinline__ T small_seq_next(const SmallParams<T>& p, SeqState<T>& seq, T gamma, double* sm_scal) {
  if (threadIdx.x < 64) {
    SeqState<T> s = seq;
    const T b = seq_next_hd<T>(p.seq_kind, p.mf, p.seq_p0, p.seq_p1, s, gamma, T(0));
    if (threadIdx.x == 0) {
      sm_scal[0] = (double)b;
      sm_scal[1] = (double)s.stepsize;
      sm_scal[2] = (double)s.theta;
      sm_scal[3] = (double)s.t;
      sm_scal[4] = (double)s.k;
    }
  }
  __syncthreads();
  seq.stepsize = (T)sm_scal[1];
  seq.theta = (T)sm_scal[2];
  seq.t = (T)sm_scal[3];
  seq.k = (long long)sm_scal[4];
  return (T)sm_scal[0];
}

// ---- back end 1: one workgroup --------------------------------------------------------------------------------
template <typename T>
struct SmallOps {
  const SmallParams<T>& p;
  double* sm_part;
  double* sm_red;
  double* sm_scal;
  T *r0, *r1, *r2;  // residual slots: 0 = p.r, 1 = rz, 2 = rz_prev (global memory)

  // a slot handle is the pointer itself (selecting among pointers by a run-time index costs scratch memory)
  using Slot = T*;
  __device__ __forceinline__ Slot slot(int i) const { return i == 0 ? r0 : (i == 1 ? r1 : r2); }
  __device__ __forceinline__ bool is_slot(Slot s, int i) const { return s == slot(i); }
  __device__ __forceinline__ T* rs(Slot s) const { return s; }
  using Vec = T*;  // an n-vector handle is the global pointer
  __device__ __forceinline__ Vec vec(int i) const { return p.buf[i]; }
  __device__ __forceinline__ double role(Vec q) const {
    for (int i = 0; i < 8; ++i)
      if (p.buf[i] == q) return (double)i;
    return -1.0;
  }
  __device__ bool aborted() const { return false; }
  __device__ bool leader() const { return true; }
  __device__ T seq_next(SeqState<T>& seq, T gamma) { return small_seq_next(p, seq, gamma, sm_scal); }
  bool tracing = false;
  __device__ void mark(int id) const {
    if (tracing && threadIdx.x == 0) p.out[32 + id] = (double)wall_clock64();
  }

  // rs[slot] = A v - b ; returns sum r^2 (to every thread)
  __device__ double residual(const T* v, Slot slot) {
    T* r_out = rs(slot);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double sq = 0.0;
    for (int rb = 0; rb < p.m; rb += 64) {
      const int i = rb + lane;
      double acc = 0.0;
      if (i < p.m) {
        for (int j = wave; j < p.n; j += SMALL_WAVES) acc += (double)p.A[i + (long long)j * p.ld] * (double)v[j];
      }
      sm_part[wave * 64 + lane] = acc;
      __syncthreads();
      if (wave == 0 && i < p.m) {
        double t = 0.0;
        for (int w = 0; w < SMALL_WAVES; ++w) t += sm_part[w * 64 + lane];
        const T ri = (T)(t - (double)p.b[i]);
        r_out[i] = ri;
        sq += (double)ri * (double)ri;
      }
      __syncthreads();
    }
    double v4[4] = {sq, 0.0, 0.0, 0.0};
    small_block_reduce<0u>(v4, sm_red);
    return v4[0];
  }

  __device__ void extrapolate(const T* a, const T* c, T beta, T* x) {  // x = a + beta (a - c)   ffb:135
    for (int j = threadIdx.x; j < p.n; j += SMALL_THREADS) x[j] = a[j] + beta * (a[j] - c[j]);
    __syncthreads();
  }

  __device__ double residual_extrap(const T* a, const T* c, T beta, T* x, Slot slot) {
    extrapolate(a, c, beta, x);
    return residual(x, slot);
  }

  // rs[so] = ca rs[sa] + cb rs[sb] ; returns sum of squares
  __device__ double residual_combo(T ca, Slot sa, T cb, Slot sb, Slot so) {
    const T *ra = rs(sa), *rb = rs(sb);
    T* ro = rs(so);
    double sq = 0.0;
    for (int i = threadIdx.x; i < p.m; i += SMALL_THREADS) {
      const T o = ca * ra[i] + cb * rb[i];
      ro[i] = o;
      sq += (double)o * (double)o;
    }
    double v4[4] = {sq, 0.0, 0.0, 0.0};
    small_block_reduce<0u>(v4, sm_red);
    return v4[0];
  }

  // g_out = lam A' rs[slot]
  __device__ void adjoint(Slot slot, T* g_out) {
    const T* r = rs(slot);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = wave; j < p.n; j += SMALL_WAVES) {
      double acc = 0.0;
      const T* col = p.A + (long long)j * p.ld;
      for (int i = lane; i < p.m; i += 64) acc += (double)col[i] * (double)r[i];
      acc = pg_wave_allreduce<false>(acc);
      if (lane == 0) g_out[j] = p.lam_ls != T(1) ? (T)(p.lam_ls * (T)acc) : (T)acc;
    }
    __syncthreads();
  }

  // y = x - gamma g ; z = prox(y) ; res = x - z ; out = { g(z), ||res||_inf, <g,res>, ||res||^2 }
  __device__ void epilogue(const T* x, const T* g, T gamma, T* y, T* z, T* res, double (&out)[4]) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const T gl = gamma * p.g_p0;
    for (int j = threadIdx.x; j < p.n; j += SMALL_THREADS) {
      T yv, zv, rv;
      small_epilogue_elem(p, x[j], g[j], gamma, gl, yv, zv, rv, acc);
      y[j] = yv;
      z[j] = zv;
      res[j] = rv;
    }
    small_block_reduce<0x2u>(acc, sm_red);
    small_epilogue_out(p, acc, out);
  }

  // [x = ea + beta (ea - ec) when ea != nullptr] ; g = lam A' rs[slot] ; epilogue
  __device__ void adjoint_epilogue(Slot slot, T* x, T* g, T gamma, T* y, T* z, T* res, double (&out)[4], bool extrap,
                                   const T* ea, const T* ec, T beta) {
    if (extrap) extrapolate(ea, ec, beta, x);
    adjoint(slot, g);
    epilogue(x, g, gamma, y, z, res, out);
  }

  __device__ void finish(Slot, Slot, Slot) {}
  __device__ double barrier_ticks() const { return 0.0; }
  __device__ double barrier_count() const { return 0.0; }
};

// ---- back end 2: W cooperating workgroups ------------------------------------------------------------------------
// Column ownership: workgroup w owns the columns [w * cols_per, ...) of A for BOTH GEMV orientations, so its slices
// of the six n-vectors never leave its LDS (loaded once, written back once); only m-vectors cross workgroups:
//   pass N   : partial_w = A[:, J_w] v[J_w]  -> global (write-through) | grid barrier | every workgroup sums the W
//              partials in the same order into its LDS copy of r (large W * m: row-sliced in two stages)
//   pass T   : g[J_w] = A[:, J_w]' r  from the LDS copy -- local
//   epilogue : local; its four scalars meet in one more grid reduction.
// A fixed-step iteration = two grid barriers and no launch; A stays L2-resident (each workgroup re-reads only its
// own columns).  Everything one workgroup writes and another reads inside the kernel goes through agent-scope (sc1,
// write-through / cache-bypassing) accesses; A and b are read-only and use ordinary cached loads.
template <typename T>
__device__ __forceinline__ T ld_ag(const T* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ void st_ag(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr long long COOP_SPIN_LIMIT = 4000000;  // polls (~1 us each) before a grid barrier gives up instead of hanging

template <typename T>
struct CoopOps {
  const SmallParams<T>& p;
  double* sm_part;
  double* sm_red;
  double* sm_scal;
  int* sm_flag;
  T* lds_base;  // [3][m_pad] residual slots, then [6][ncol_pad] slices of the n-vectors
  int c0, nc;   // this workgroup's columns [c0, c0 + nc)
  unsigned long long bar_target;
  int npass, nred;
  bool dead;
  long long t_bar = 0, n_bar = 0;  // telemetry: 100 MHz ticks spent inside grid barriers, number of barriers

  using Slot = int;  // index of an m_pad-sized LDS region
  __device__ __forceinline__ Slot slot(int i) const { return i; }
  __device__ __forceinline__ bool is_slot(Slot s, int i) const { return s == i; }
  __device__ __forceinline__ T* rs(Slot s) const { return lds_base + (size_t)s * p.m_pad; }
  using Vec = int;  // index of an ncol_pad-sized LDS region (this workgroup's slice of an n-vector)
  __device__ __forceinline__ Vec vec(int i) const { return i; }
  __device__ __forceinline__ double role(Vec v) const { return (double)v; }
  __device__ __forceinline__ T* vs(Vec v) const { return lds_base + (size_t)3 * p.m_pad + (size_t)v * p.ncol_pad; }
  __device__ bool aborted() const { return dead; }
  __device__ bool leader() const { return blockIdx.x == 0; }
  __device__ T seq_next(SeqState<T>& seq, T gamma) { return small_seq_next(p, seq, gamma, sm_scal); }
  bool tracing = false;
  __device__ void mark(int id) const {
    if (tracing && blockIdx.x == 0 && threadIdx.x == 0) p.out[32 + id] = (double)wall_clock64();
  }

  __device__ void load_state() {
    for (int h = 0; h < 6; ++h) {
      T* dst = vs(h);
      for (int j = threadIdx.x; j < nc; j += SMALL_THREADS) dst[j] = p.buf[h][c0 + j];
    }
    if (p.reuse) {  // the line-search residual pair continues from the host-driven steps
      T *d1 = rs(1), *d2 = rs(2);
      for (int i = threadIdx.x; i < p.m; i += SMALL_THREADS) {
        d1[i] = p.buf[6][i];
        d2[i] = p.buf[7][i];
      }
    }
    __syncthreads();
  }

  // Grid barrier: every thread drains its write-through stores, the workgroup's thread 0 takes a ticket on a
  // monotonically increasing counter and polls it.  Bounded: on a timeout the solve is abandoned (flag), never hung.
  __device__ void barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (p.W == 1 || dead) return;
#ifdef PG_COOP_TRACE
    const long long t_in = wall_clock64();
#endif
    bar_target += (unsigned long long)p.W;
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(p.bar, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int ok = 1;
      long long spins = 0;
      while (__hip_atomic_load(p.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < bar_target) {
        __builtin_amdgcn_s_sleep(1);
        ++spins;
        if (spins > COOP_SPIN_LIMIT ||
            ((spins & 4095) == 0 && __hip_atomic_load(p.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
          ok = 0;
          break;
        }
      }
      if (!ok) __hip_atomic_store(p.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *sm_flag = ok;
    }
    __syncthreads();
    if (*sm_flag == 0) dead = true;
#ifdef PG_COOP_TRACE
    t_bar += wall_clock64() - t_in;
    n_bar += 1;
#endif
  }

  // grid-wide reduction of 4 doubles: block partial -> global slot -> barrier -> every workgroup sums the W partials in
  // the same order (bit-identical results everywhere)
  template <unsigned MAXMASK>
  __device__ void grid_reduce(double (&v)[4]) {
    small_block_reduce<MAXMASK>(v, sm_red);
    PG_MARK(*this, 18);
    if (p.W == 1) return;
    double* slot = p.spart + (size_t)(nred & 1) * p.W * 4;
    if (threadIdx.x < 4) st_ag(slot + (size_t)blockIdx.x * 4 + threadIdx.x, v[threadIdx.x]);
    ++nred;
    barrier();
    PG_MARK(*this, 19);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = ((MAXMASK >> k) & 1u) ? -INFINITY : 0.0;
    if (dead) return;
    for (int w = threadIdx.x; w < p.W; w += SMALL_THREADS) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double o = ld_ag(slot + (size_t)w * 4 + k);
        v[k] = ((MAXMASK >> k) & 1u) ? pg_maxn(v[k], o) : (v[k] + o);
      }
    }
    PG_MARK(*this, 20);
    small_block_reduce<MAXMASK>(v, sm_red);
    PG_MARK(*this, 21);
  }

  // pass N over the own columns: partial_w[i] = sum_j A[i, c0 + j] v[j]   (v: LDS slice)
  __device__ void pass_n(const T* v) {
    PG_MARK(*this, 10);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* part = p.npart + ((size_t)(npass & 1) * p.W + blockIdx.x) * p.m_pad;
    const T* Aj = p.A + (long long)c0 * p.ld;
    if (p.nrb >= SMALL_WAVES) {  // a wave per 64-row block, all own columns
      for (int rb = wave; rb < p.nrb; rb += SMALL_WAVES) {
        const int i = rb * 64 + lane;
        double acc = 0.0;
        if (i < p.m)
          for (int j = 0; j < nc; ++j) acc += (double)Aj[i + (long long)j * p.ld] * (double)v[j];
        st_ag(part + i, acc);
      }
    } else {  // few row blocks: G waves share one, each taking every G-th column; fixed-order combine through LDS
      const int G = SMALL_WAVES / p.nrb;
      const int rb = wave / G, cg = wave - rb * G;
      const int i = rb * 64 + lane;
      double acc = 0.0;
      if (rb < p.nrb && i < p.m)
        for (int j = cg; j < nc; j += G) acc += (double)Aj[i + (long long)j * p.ld] * (double)v[j];
      sm_part[wave * 64 + lane] = acc;
      __syncthreads();
      if (rb < p.nrb && cg == 0) {
        double t = 0.0;
        for (int q = 0; q < G; ++q) t += sm_part[(wave + q) * 64 + lane];
        st_ag(part + i, t);
      }
    }
    ++npass;
    PG_MARK(*this, 11);
    barrier();
    PG_MARK(*this, 12);
  }

  // sum over the workgroups w0 <= w < w1 of partial_w[i], four loads in flight
  __device__ __forceinline__ double sum_partials(const double* part, int i, int w0, int w1) const {
    double t = 0.0;
    int w = w0;
    for (; w + 4 <= w1; w += 4) {
      const double a0 = ld_ag(part + (size_t)w * p.m_pad + i), a1 = ld_ag(part + (size_t)(w + 1) * p.m_pad + i),
                   a2 = ld_ag(part + (size_t)(w + 2) * p.m_pad + i), a3 = ld_ag(part + (size_t)(w + 3) * p.m_pad + i);
      t += a0;
      t += a1;
      t += a2;
      t += a3;
    }
    for (; w < w1; ++w) t += ld_ag(part + (size_t)w * p.m_pad + i);
    return t;
  }

  // rs(slot) = sum_w partial_w - b in every workgroup (same order everywhere) ; returns sum r^2
  __device__ double combine(Slot slot) {
    const double* part = p.npart + (size_t)((npass - 1) & 1) * p.W * p.m_pad;
    T* dst = rs(slot);
    double sq = 0.0;
    if (p.two_stage) {  // stage 1: each workgroup sums its own rows and publishes them; stage 2: everybody reads r
      T* rf = p.rfull + (size_t)((npass - 1) & 1) * p.m_pad;
      if (!dead) {
        const int i0 = blockIdx.x * p.rows_per;
        const int i1 = (i0 + p.rows_per < p.m) ? (i0 + p.rows_per) : p.m;
        for (int i = i0 + (int)threadIdx.x; i < i1; i += SMALL_THREADS)
          st_ag(rf + i, (T)(sum_partials(part, i, 0, p.W) - (double)p.b[i]));
      }
      barrier();
      if (!dead)
        for (int i = threadIdx.x; i < p.m; i += SMALL_THREADS) {
          const T ri = ld_ag(rf + i);
          dst[i] = ri;
          sq += (double)ri * (double)ri;
        }
    } else if (p.m_pad * 2 <= SMALL_THREADS) {
      // few rows: Q threads share a row, each summing a contiguous segment of the workgroups; the Q segment sums are
      // added in segment order through LDS (same order in every workgroup)
      const int Q = SMALL_THREADS / p.m_pad;  // 2 .. 16, Q * m_pad <= 1024 = the size of sm_part
      const int q = (int)threadIdx.x / p.m_pad, i = (int)threadIdx.x - q * p.m_pad;
      const int seg = (p.W + Q - 1) / Q;
      if (!dead && q < Q && i < p.m) {
        const int w0 = q * seg, w1 = (w0 + seg < p.W) ? (w0 + seg) : p.W;
        sm_part[q * p.m_pad + i] = w0 < w1 ? sum_partials(part, i, w0, w1) : 0.0;
      }
      __syncthreads();
      if (!dead && q == 0 && i < p.m) {
        double t = sm_part[i];
        for (int qq = 1; qq < Q; ++qq) t += sm_part[qq * p.m_pad + i];
        const T ri = (T)(t - (double)p.b[i]);
        dst[i] = ri;
        sq = (double)ri * (double)ri;
      }
    } else if (!dead) {
      for (int i = threadIdx.x; i < p.m; i += SMALL_THREADS) {
        const T ri = (T)(sum_partials(part, i, 0, p.W) - (double)p.b[i]);
        dst[i] = ri;
        sq += (double)ri * (double)ri;
      }
    }
    PG_MARK(*this, 13);
    double v4[4] = {sq, 0.0, 0.0, 0.0};
    small_block_reduce<0u>(v4, sm_red);  // ends with workgroup barriers: rs(slot) is complete
    PG_MARK(*this, 14);
    return v4[0];
  }

  __device__ double residual(Vec v, Slot slot) {
    pass_n(vs(v));
    return combine(slot);
  }

  __device__ void extrapolate(Vec a, Vec c, T beta, Vec x) {  // x = a + beta (a - c) on the own columns   ffb:135
    const T *av = vs(a), *cv = vs(c);
    T* xv = vs(x);
    for (int j = threadIdx.x; j < nc; j += SMALL_THREADS) xv[j] = av[j] + beta * (av[j] - cv[j]);
    __syncthreads();
    PG_MARK(*this, 15);
  }

  __device__ double residual_extrap(Vec a, Vec c, T beta, Vec x, Slot slot) {
    extrapolate(a, c, beta, x);
    return residual(x, slot);
  }

  __device__ double residual_combo(T ca, Slot sa, T cb, Slot sb, Slot so) {  // redundantly in every workgroup (LDS)
    const T *ra = rs(sa), *rb = rs(sb);
    T* ro = rs(so);
    double sq = 0.0;
    for (int i = threadIdx.x; i < p.m; i += SMALL_THREADS) {
      const T o = ca * ra[i] + cb * rb[i];
      ro[i] = o;
      sq += (double)o * (double)o;
    }
    double v4[4] = {sq, 0.0, 0.0, 0.0};
    small_block_reduce<0u>(v4, sm_red);
    return v4[0];
  }

  // g[J_w] = lam A[:, J_w]' rs(slot): a wave per own column, no cross-workgroup traffic
  __device__ void adjoint_local(Slot slot, Vec g) {
    const T* r = rs(slot);
    T* gv = vs(g);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = wave; j < nc; j += SMALL_WAVES) {
      double a = 0.0;
      const T* col = p.A + (long long)(c0 + j) * p.ld;
      for (int i = lane; i < p.m; i += 64) a += (double)col[i] * (double)r[i];
      a = pg_wave_allreduce<false>(a);
      if (lane == 0) gv[j] = p.lam_ls != T(1) ? (T)(p.lam_ls * (T)a) : (T)a;
    }
    __syncthreads();
    PG_MARK(*this, 16);
  }

  __device__ void adjoint(Slot slot, Vec g) { adjoint_local(slot, g); }

  __device__ void epilogue(Vec x, Vec g, T gamma, Vec y, Vec z, Vec res, double (&out)[4]) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const T gl = gamma * p.g_p0;
    const T *xv = vs(x), *gv = vs(g);
    T *yv = vs(y), *zv = vs(z), *rv = vs(res);
    for (int j = threadIdx.x; j < nc; j += SMALL_THREADS) {
      T yo, zo, ro;
      small_epilogue_elem(p, xv[j], gv[j], gamma, gl, yo, zo, ro, acc);
      yv[j] = yo;
      zv[j] = zo;
      rv[j] = ro;
    }
    PG_MARK(*this, 17);
    grid_reduce<0x2u>(acc);
    small_epilogue_out(p, acc, out);
  }

  __device__ void adjoint_epilogue(Slot slot, Vec x, Vec g, T gamma, Vec y, Vec z, Vec res, double (&out)[4],
                                   bool extrap, Vec ea, Vec ec, T beta) {
    if (extrap) extrapolate(ea, ec, beta, x);
    adjoint_local(slot, g);
    epilogue(x, g, gamma, y, z, res, out);
  }

  __device__ double barrier_ticks() const { return (double)t_bar; }
  __device__ double barrier_count() const { return (double)n_bar; }

  // write the LDS-resident state back: own slices of the six n-vectors; workgroup 0 also exports the residuals
  __device__ void finish(Slot s_r, Slot, Slot) {
    __syncthreads();
    for (int h = 0; h < 6; ++h) {
      const T* src = vs(h);
      for (int j = threadIdx.x; j < nc; j += SMALL_THREADS) p.buf[h][c0 + j] = src[j];
    }
    if (blockIdx.x != 0) return;
    const T *src = rs(s_r), *s1 = rs(1), *s2 = rs(2);
    for (int i = threadIdx.x; i < p.m; i += SMALL_THREADS) {
      p.r[i] = src[i];
      if (p.reuse) {
        p.buf[6][i] = s1[i];  // slot-wise: the roles travel in the result block
        p.buf[7][i] = s2[i];
      }
    }
  }
};

// ---- the loop (both back ends) -------------------------------------------------------------------------------------
template <typename T, typename Ops>
__device__ void solver_loop(const SmallParams<T>& p, Ops& ops) {
  using Vec = typename Ops::Vec;
  Vec x = ops.vec(0), grad = ops.vec(1), y = ops.vec(2), z = ops.vec(3), res = ops.vec(4), zp = ops.vec(5);
  typename Ops::Slot s_r = ops.slot(0), s_rz = ops.slot(1), s_rzp = ops.slot(2);  // scratch r, A z - b, A z_prev - b
  T gamma = p.gamma, f_x = p.f_x, g_z = p.g_z, res_inf = p.res_inf, dot_gr = p.dot_gr, res_sq = p.res_sq;
  T beta = T(0);
  SeqState<T> seq = p.seq;
  long long k = p.k_start, nbt_total = 0, passes = 0;
  int flags = 0;
  bool rz_valid = false;
#ifdef PG_COOP_TRACE
  const long long t_begin = wall_clock64();
#endif
  const T eps = sizeof(T) == 4 ? (T)1.1920928955078125e-07 : (T)2.220446049250313e-16;
  const T f_scale = (T)0.5 * p.lam_ls;
  double e4[4];

  auto model = [&]() -> T {  // fb_tools.jl:3-5 with L = 1 / gamma
    const T L = T(1) / gamma;
    return f_x - dot_gr + (L / T(2)) * res_sq;
  };
  auto set_epilogue = [&]() {
    g_z = (T)e4[0];
    res_inf = e4[2] != e4[2] ? (T)e4[2] : (T)e4[1];  // (a NaN <grad, res> is not a converged state: pg_iter.hip::res_inf_guarded)
    dot_gr = (T)e4[2];
    res_sq = (T)e4[3];
  };

  while (!ops.aborted() && !(k >= p.maxit || res_inf / gamma <= p.tol)) {  // ProximalAlgorithms.jl:117 ; fb:125-126
#ifdef PG_COOP_TRACE
    ops.tracing = (k == p.k_start + 3);
#endif
    PG_MARK(ops, 0);
    if (p.fast) {
      if (p.adaptive) {  // fast_forward_backward.jl:110-129 + fb_tools.jl:24-63
        gamma = gamma * p.increase_gamma;
        T f_upp = model();
        const typename Ops::Slot s_dst = p.reuse ? s_rz : s_r;  // without the residual pair the line search uses the scratch slot
        T f_z = f_scale * (T)ops.residual(z, s_dst);
        passes += 1;
        T tol_ls = T(10) * eps * (T(1) + fabs(f_z));
        while (f_z > f_upp + tol_ls && gamma >= p.min_gamma && !ops.aborted()) {
          gamma = gamma * p.reduce_gamma;
          ops.epilogue(x, grad, gamma, y, z, res, e4);
          set_epilogue();
          f_upp = model();
          f_z = f_scale * (T)ops.residual(z, s_dst);
          passes += 1;
          tol_ls = T(10) * eps * (T(1) + fabs(f_z));
          nbt_total += 1;
        }
        if (gamma < p.min_gamma) flags |= PG_FLAG_GAMMA_TOO_SMALL;
        rz_valid = true;
      } else if (p.has_fixed_gamma) {
        gamma = p.fixed_gamma;  // :131
      }
      PG_MARK(ops, 1);
      beta = ops.seq_next(seq, gamma);  // :134
      PG_MARK(ops, 2);
      {  // :136 (the extrapolation :135 is formed below from the swapped pair: x = zp + beta (zp - z))
        Vec t = zp;
        zp = z;
        z = t;
      }
      if (p.adaptive && p.reuse && rz_valid) {  // A x - b = (1 + beta)(A z - b) - beta (A z_prev - b)
        f_x = (T)((double)f_scale * ops.residual_combo(T(1) + beta, s_rz, -beta, s_rzp, s_r));
        const typename Ops::Slot t = s_rzp;
        s_rzp = s_rz;
        s_rz = t;
        rz_valid = false;
        ops.adjoint_epilogue(s_r, x, grad, gamma, y, z, res, e4, true, zp, z, beta);  // :135, :138-142
      } else {
        f_x = f_scale * (T)ops.residual_extrap(zp, z, beta, x, s_r);  // :135, :138
        PG_MARK(ops, 3);
        passes += 1;
        rz_valid = false;  // this residual belongs to x, not to z: nothing to reuse next time
        ops.adjoint_epilogue(s_r, x, grad, gamma, y, z, res, e4, false, x, x, T(0));  // :138-142
      }
      passes += 1;
      set_epilogue();
      PG_MARK(ops, 4);
    } else {
      if (p.adaptive) {  // forward_backward.jl:90-110 ; gradient at z is kept (zp plays grad_f_z)
        gamma = gamma * p.increase_gamma;
        T f_upp = model();
        T f_z = f_scale * (T)ops.residual(z, s_r);
        ops.adjoint(s_r, zp);
        passes += 2;
        T tol_ls = T(10) * eps * (T(1) + fabs(f_z));
        while (f_z > f_upp + tol_ls && gamma >= p.min_gamma && !ops.aborted()) {
          gamma = gamma * p.reduce_gamma;
          ops.epilogue(x, grad, gamma, y, z, res, e4);
          set_epilogue();
          f_upp = model();
          f_z = f_scale * (T)ops.residual(z, s_r);
          ops.adjoint(s_r, zp);
          passes += 2;
          tol_ls = T(10) * eps * (T(1) + fabs(f_z));
          nbt_total += 1;
        }
        if (gamma < p.min_gamma) flags |= PG_FLAG_GAMMA_TOO_SMALL;
        f_x = f_z;  // :92
        Vec t = x;   // :109
        x = z;
        z = t;
        t = grad;   // :110
        grad = zp;
        zp = t;
        ops.epilogue(x, grad, gamma, y, z, res, e4);  // :117-120
      } else {  // :111-115
        Vec t = x;
        x = z;
        z = t;
        f_x = f_scale * (T)ops.residual(x, s_r);
        passes += 2;
        ops.adjoint_epilogue(s_r, x, grad, gamma, y, z, res, e4, false, x, x, T(0));  // :113-120
      }
      set_epilogue();
    }
    ++k;
  }
  ops.finish(s_r, s_rz, s_rzp);
  if (ops.leader() && threadIdx.x == 0) {
    double* o = p.out;
    o[0] = (double)k;
    o[1] = (double)gamma;
    o[2] = (double)f_x;
    o[3] = (double)g_z;
    o[4] = (double)res_inf;
    o[5] = (double)dot_gr;
    o[6] = (double)res_sq;
    o[7] = (double)beta;
    o[8] = (double)seq.stepsize;
    o[9] = (double)seq.theta;
    o[10] = (double)seq.t;
    o[11] = (double)seq.k;
    o[12] = ops.role(x);
    o[13] = ops.role(grad);
    o[14] = ops.role(y);
    o[15] = ops.role(z);
    o[16] = ops.role(res);
    o[17] = ops.role(zp);
    o[18] = ops.is_slot(s_rz, 1) ? 6.0 : 7.0;
    o[19] = ops.is_slot(s_rzp, 2) ? 7.0 : 6.0;
    o[20] = (double)nbt_total;
    o[21] = (double)flags;
    o[22] = (double)passes;
    o[23] = rz_valid ? 1.0 : 0.0;
    o[24] = ops.aborted() ? 1.0 : 0.0;
#ifdef PG_COOP_TRACE
    o[25] = (double)(wall_clock64() - t_begin);  // telemetry (100 MHz ticks): whole loop, inside grid barriers, count
    o[26] = ops.barrier_ticks();
    o[27] = ops.barrier_count();
#else
    o[25] = o[26] = o[27] = 0.0;
#endif
  }
}

template <typename T>
__global__ __launch_bounds__(SMALL_THREADS) void small_solver_kernel(SmallParams<T> p) {
  __shared__ double sm_part[SMALL_WAVES * 64];
  __shared__ double sm_red[SMALL_WAVES * 4];
  __shared__ double sm_scal[8];
  SmallOps<T> ops{p, sm_part, sm_red, sm_scal, p.r, p.buf[6], p.buf[7]};
  solver_loop<T, SmallOps<T>>(p, ops);
}

template <typename T>
__global__ __launch_bounds__(SMALL_THREADS) void coop_solver_kernel(SmallParams<T> p) {
  __shared__ double sm_part[SMALL_WAVES * 64];
  __shared__ double sm_red[SMALL_WAVES * 4];
  __shared__ double sm_scal[8];
  __shared__ int sm_flag;
  extern __shared__ __attribute__((aligned(16))) unsigned char coop_lds[];
  const int c0 = (int)blockIdx.x * p.cols_per;
  int nc = p.n - c0;
  if (nc > p.cols_per) nc = p.cols_per;
  if (nc < 0) nc = 0;
  CoopOps<T> ops{p, sm_part, sm_red, sm_scal, &sm_flag, reinterpret_cast<T*>(coop_lds), c0, nc, 0ull, 0, 0, false};
  ops.load_state();
  solver_loop<T, CoopOps<T>>(p, ops);
}

template <typename T>
void small_fill_params(pg_iter* it, SmallParams<T>& p, void* (&bufs)[8], int64_t k_start, int64_t maxit, double tol) {
  pg_mat* A = it->f->A;
  memset(&p, 0, sizeof(p));
  p.A = (const T*)A->data;
  p.ld = A->ld;
  p.m = (int)A->m;
  p.n = (int)A->n;
  p.b = (const T*)it->f->b;
  void* roles[8] = {it->x, it->grad_f_x, it->y, it->z, it->res, it->o.fast ? it->z_prev : it->grad_f_z, it->rz, it->rz_prev};
  for (int i = 0; i < 8; ++i) {
    bufs[i] = roles[i];
    p.buf[i] = (T*)roles[i];
  }
  p.r = (T*)it->f->r;
  p.fast = it->o.fast;
  p.adaptive = it->adaptive ? 1 : 0;
  p.reuse = (it->rz != nullptr && it->rz_prev != nullptr) ? 1 : 0;
  p.g_kind = it->o.g_kind;
  p.seq_kind = it->o.seq_kind;
  p.has_fixed_gamma = (it->o.gamma > 0 || it->o.Lf > 0) ? 1 : 0;
  p.fixed_gamma = p.has_fixed_gamma ? (T)(it->o.gamma > 0 ? it->o.gamma : (double)(T(1) / (T)it->o.Lf)) : T(0);
  p.g_p0 = (T)it->o.g_p0;
  p.g_p1 = (T)it->o.g_p1;
  p.lam_ls = (T)it->f->lam;
  p.gamma = (T)it->gamma;
  p.f_x = (T)it->f_x;
  p.g_z = (T)it->g_z;
  p.res_inf = (T)it->res_inf;
  p.dot_gr = (T)it->dot_gr;
  p.res_sq = (T)it->res_sq;
  p.min_gamma = (T)it->o.minimum_gamma;
  p.reduce_gamma = (T)it->o.reduce_gamma;
  p.increase_gamma = (T)it->o.increase_gamma;
  p.mf = (T)it->o.mf;
  p.seq_p0 = (T)it->o.seq_p0;
  p.seq_p1 = (T)it->o.seq_p1;
  p.seq = SeqState<T>{(T)it->seq_stepsize, (T)it->seq_theta, (T)it->seq_t, (long long)it->seq_k};
  p.k_start = k_start;
  p.maxit = maxit;
  p.tol = (T)tol;
  p.W = 1;
}

pg_status small_result_block(pg_ctx* c) {
  if (!c->small_out) {
    double* host = nullptr;
    PG_HIP(hipHostMalloc((void**)&host, sizeof(double) * SMALL_OUT_DOUBLES, hipHostMallocMapped));
    c->small_out_host = host;
    PG_HIP(hipHostGetDevicePointer((void**)&c->small_out, host, 0));
  }
  return PG_OK;
}

// adopt the state the persistent kernel left behind (scalars, buffer roles, telemetry)
void small_read_result(pg_iter* it, void* (&bufs)[8], int64_t* k_out) {
  const double* o = it->ctx->small_out_host;
  *k_out = (int64_t)o[0];
  it->gamma = o[1];
  it->f_x = o[2];
  it->g_z = o[3];
  it->res_inf = o[4];
  it->dot_gr = o[5];
  it->res_sq = o[6];
  it->beta = o[7];
  it->seq_stepsize = o[8];
  it->seq_theta = o[9];
  it->seq_t = o[10];
  it->seq_k = (int64_t)o[11];
  auto at = [&](int idx) -> void* { return (idx >= 0 && idx < 8) ? bufs[idx] : nullptr; };
  it->x = at((int)o[12]);
  it->grad_f_x = at((int)o[13]);
  it->y = at((int)o[14]);
  it->z = at((int)o[15]);
  it->res = at((int)o[16]);
  if (it->o.fast)
    it->z_prev = at((int)o[17]);
  else
    it->grad_f_z = at((int)o[17]);
  if (it->rz != nullptr && it->rz_prev != nullptr) {
    it->rz = at((int)o[18]);
    it->rz_prev = at((int)o[19]);
  }
  it->n_backtracks = (int)o[20];
  it->flags = (int)o[21];
  it->f->a_passes += (int64_t)o[22];
  it->rz_valid = false;  // the next host-driven step evaluates A x itself
  it->sp_ready = false;  // ... and any single-sweep speculation is gone
  it->f_z = it->f_z_upp = NAN;
}

template <typename T>
pg_status iter_run_small(pg_iter* it, int64_t k_start, int64_t maxit, double tol, int64_t* k_out) {
  pg_ctx* c = it->ctx;
  SmallParams<T> p;
  void* bufs[8];
  small_fill_params<T>(it, p, bufs, k_start, maxit, tol);
  PG_TRY(small_result_block(c));
  p.out = c->small_out;
  hipLaunchKernelGGL(small_solver_kernel<T>, dim3(1), dim3(SMALL_THREADS), 0, c->stream, p);
  PG_LAUNCH_CHECK();
  PG_HIP(hipStreamSynchronize(c->stream));
  small_read_result(it, bufs, k_out);
  return PG_OK;
}

// cooperative multi-workgroup variant; blocks <= 0: chosen from the size of A
constexpr int64_t COOP_MAX_LDS = 128 * 1024;  // three residual vectors + six n-vector slices per workgroup

template <typename T>
pg_status iter_run_coop(pg_iter* it, int64_t k_start, int64_t maxit, double tol, int blocks, int64_t* k_out) {
  pg_ctx* c = it->ctx;
  pg_mat* A = it->f->A;
  SmallParams<T> p;
  void* bufs[8];
  small_fill_params<T>(it, p, bufs, k_start, maxit, tol);
  PG_TRY(small_result_block(c));
  p.out = c->small_out;
  p.m_pad = (int)pg_round_up(A->m, 64);
  p.nrb = p.m_pad / 64;
  int W = blocks;
  if (W <= 0) {
    // barrier and combination costs grow with the grid, the per-workgroup GEMV work shrinks with it: measured optima
    // sit near sqrt(bytes of A / 1 KiB) workgroups (28 for 200x500 f64, ~40 for 500x1000 f64)
    const double kib = (double)(A->m * A->n * (int64_t)sizeof(T)) / 1024.0;
    W = (int)std::lround(std::sqrt(kib));
  }
  if (W > c->num_cu) W = c->num_cu;  // one workgroup per CU: all of them are resident, the grid barrier cannot starve
  if ((int64_t)W > A->n) W = (int)A->n;
  if (W < 1) W = 1;
  // every workgroup keeps 3 m-vectors and its slice of 6 n-vectors in LDS: widen the grid until the slices fit
  auto lds_for = [&](int w) -> int64_t {
    const int64_t cols = (A->n + w - 1) / w;
    return ((int64_t)3 * p.m_pad + 6 * pg_round_up(cols, 16)) * (int64_t)sizeof(T);
  };
  while (lds_for(W) > COOP_MAX_LDS && W < c->num_cu) ++W;
  if (lds_for(W) > COOP_MAX_LDS) {
    pg_set_error("the cooperative solver needs %lld bytes of LDS per workgroup for this shape (limit %lld); use pg_iter_run",
                 (long long)lds_for(W), (long long)COOP_MAX_LDS);
    return PG_ERR_UNSUPPORTED;
  }
  p.cols_per = (int)((A->n + W - 1) / W);
  W = (int)((A->n + p.cols_per - 1) / p.cols_per);  // no workgroup without columns
  p.ncol_pad = (int)pg_round_up(p.cols_per, 16);
  p.W = W;
  p.two_stage = ((int64_t)W * A->m > 65536) ? 1 : 0;
  p.rows_per = (int)((A->m + W - 1) / W);
  const size_t lds = (size_t)lds_for(W);
  const void* kern = reinterpret_cast<const void*>(&coop_solver_kernel<T>);
  {
    static std::mutex mu;
    static bool opted_in[64] = {};
    std::lock_guard<std::mutex> lock(mu);
    if (!opted_in[c->device & 63]) {
      PG_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)COOP_MAX_LDS));
      opted_in[c->device & 63] = true;
    }
  }
  // workspace: [bar (8 B) | abort (4 B) | pad to 64 B][2][W][4] scalar partials [2][W][m_pad] pass-N partials [2][m_pad] r
  const size_t n_sp = (size_t)2 * W * 4, n_np = (size_t)2 * W * p.m_pad;
  const size_t need = 64 + sizeof(double) * (n_sp + n_np) + sizeof(T) * (size_t)2 * p.m_pad;
  if (c->coop_ws_bytes < need) {
    if (c->coop_ws) {
      PG_HIP(hipStreamSynchronize(c->stream));
      (void)hipFree(c->coop_ws);
      c->coop_ws = nullptr;
      c->coop_ws_bytes = 0;
    }
    hipError_t e = hipMalloc(&c->coop_ws, need);
    if (e != hipSuccess) {
      pg_set_error("hipMalloc for the cooperative solver workspace failed: %s", hipGetErrorString(e));
      return PG_ERR_ALLOC;
    }
    c->coop_ws_bytes = need;
  }
  char* ws = (char*)c->coop_ws;
  PG_HIP(hipMemsetAsync(ws, 0, 64, c->stream));
  p.bar = (unsigned long long*)ws;
  p.abort_flag = (int*)(ws + 8);
  p.spart = (double*)(ws + 64);
  p.npart = p.spart + n_sp;
  p.rfull = (T*)(p.npart + n_np);
  void* args[1] = {(void*)&p};
  hipError_t e;
  {
    std::lock_guard<std::mutex> lock(pg_coop_launch_mutex());  // (see pg_internal.h)
    e = hipLaunchCooperativeKernel(kern, dim3((unsigned)W), dim3(SMALL_THREADS), args, (unsigned)lds, c->stream);
  }
  if (e != hipSuccess) {
    pg_set_error("cooperative launch (%d workgroups, %zu bytes of LDS) failed: %s", W, lds, hipGetErrorString(e));
    return PG_ERR_HIP;
  }
  PG_HIP(hipStreamSynchronize(c->stream));
  small_read_result(it, bufs, k_out);
#ifdef PG_COOP_TRACE
  if (getenv("PG_COOP_VERBOSE")) {
    const double* o = c->small_out_host;
    const double its = (double)(*k_out - k_start);
    fprintf(stderr, "[pg coop] %dx%d W=%d cols/wg=%d two_stage=%d: %.0f iterations, %.2f us/iteration, %.1f barriers/iteration, %.2f us/barrier (%.0f %% of the loop)\n",
            p.m, p.n, W, p.cols_per, p.two_stage, its, o[25] * 0.01 / (its > 0 ? its : 1), o[27] / (its > 0 ? its : 1),
            o[27] > 0 ? o[26] * 0.01 / o[27] : 0.0, o[25] > 0 ? 100.0 * o[26] / o[25] : 0.0);
  }
  {
    const double* o = c->small_out_host + 32;
    const char* names[22] = {"loop top", "before seq", "after seq", "after residual_extrap", "after adjoint_epilogue", "", "", "", "", "",
                             "pass_n entry", "pass_n stored", "pass_n barrier done", "combine loaded", "combine reduced", "extrapolated",
                             "adjoint done", "epilogue local done", "reduce: block", "reduce: barrier done", "reduce: loaded", "reduce: done"};
    const int order[] = {0, 1, 2, 15, 10, 11, 12, 13, 14, 3, 16, 17, 18, 19, 20, 21, 4};
    double prev = o[0];
    for (int q : order) {
      fprintf(stderr, "[pg coop trace] %-24s +%7.2f us (t = %8.2f)\n", names[q], (o[q] - prev) * 0.01, (o[q] - o[0]) * 0.01);
      prev = o[q];
    }
  }
#endif
  if (c->small_out_host[24] != 0.0) {
    pg_set_error("the cooperative solver gave up at a grid barrier (a workgroup did not arrive)");
    return PG_ERR_HIP;
  }
  return PG_OK;
}

}  // namespace

extern "C" {

// Whole solve in one launch of one workgroup (launch-bound sizes: m * n <= 2^20 elements).
pg_status pg_iter_run_small(pg_iter* it, int64_t k_start, int64_t maxit, double tol, int64_t* k_out,
                            pg_iter_scalars* out) {
  PG_REQUIRE(it != nullptr && k_out != nullptr, "null argument");
  PG_REQUIRE(it->initialized, "pg_iter_init has not been called");
  PG_REQUIRE(it->o.seq_kind != PG_SEQ_HOST || !it->o.fast, "PG_SEQ_HOST needs per-step coefficients");
  PG_REQUIRE(it->ctx->allreduce == nullptr && it->ctx->allreduce_begin == nullptr,
             "the single-workgroup solver does not support row-sharded operators");
  if (it->g_v0 != nullptr) {
    pg_set_error("the single-workgroup solver takes scalar parameters of g only (per-element bounds / weights: pg_iter_run)");
    return PG_ERR_UNSUPPORTED;
  }
  pg_mat* A = it->f->A;
  if (A->m * A->n > ((int64_t)1 << 20) || A->m >= ((int64_t)1 << 31) || A->n >= ((int64_t)1 << 31) || A->m == 0 || A->n == 0) {
    pg_set_error("pg_iter_run_small is for launch-bound sizes (0 < m * n <= 2^20 elements); use pg_iter_run");
    return PG_ERR_UNSUPPORTED;
  }
  PG_TRY(it->dtype == PG_F32 ? iter_run_small<float>(it, k_start, maxit, tol, k_out)
                             : iter_run_small<double>(it, k_start, maxit, tol, k_out));
  fill_scalars(it, out);
  return PG_OK;
}

// Whole solve in one cooperative launch of up to one workgroup per CU that meet at grid barriers (sizes between the
// single-workgroup solver and the streaming kernels: A stays cache-resident, three m-vectors live in LDS).
pg_status pg_iter_run_coop(pg_iter* it, int64_t k_start, int64_t maxit, double tol, int32_t blocks, int64_t* k_out,
                           pg_iter_scalars* out) {
  PG_REQUIRE(it != nullptr && k_out != nullptr, "null argument");
  PG_REQUIRE(it->initialized, "pg_iter_init has not been called");
  PG_REQUIRE(it->o.seq_kind != PG_SEQ_HOST || !it->o.fast, "PG_SEQ_HOST needs per-step coefficients");
  PG_REQUIRE(it->ctx->allreduce == nullptr && it->ctx->allreduce_begin == nullptr,
             "the cooperative solver does not support row-sharded operators");
  if (it->g_v0 != nullptr) {
    pg_set_error("the cooperative solver takes scalar parameters of g only (per-element bounds / weights: pg_iter_run)");
    return PG_ERR_UNSUPPORTED;
  }
  pg_mat* A = it->f->A;
  const int64_t es = (int64_t)pg_sizeof(it->dtype);
  if (A->m == 0 || A->n == 0 || A->n >= ((int64_t)1 << 31) || 3 * pg_round_up(A->m, 64) * es > COOP_MAX_LDS * 3 / 4 ||
      A->m * A->n * es > ((int64_t)256 << 20)) {
    pg_set_error("pg_iter_run_coop needs 0 < m <= %lld rows (three residual vectors in LDS) and at most 256 MiB of A; "
                 "use pg_iter_run", (long long)(COOP_MAX_LDS / (4 * es)));
    return PG_ERR_UNSUPPORTED;
  }
  PG_TRY(it->dtype == PG_F32 ? iter_run_coop<float>(it, k_start, maxit, tol, blocks, k_out)
                             : iter_run_coop<double>(it, k_start, maxit, tol, blocks, k_out));
  fill_scalars(it, out);
  return PG_OK;
}

}  // extern "C"
