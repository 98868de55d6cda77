// Internal declarations shared by pg_iter.hip (host-driven iterations) and pg_persist.hip (persistent solver kernels).
#pragma once
#include <cmath>
#include <cstdint>

#include "pg_internal.h"

struct pg_iter {
  pg_ctx* ctx = nullptr;
  pg_ls* f = nullptr;
  pg_iter_opts o{};
  const void *g_v0 = nullptr, *g_v1 = nullptr;  // IndBox: per-element bounds; NormL1: weights in g_v0 (pg_iter_set_g_vectors), borrowed
  int dtype = PG_F32;
  int64_t n = 0;
  void* slab = nullptr;  // one allocation holding all state vectors
  // state vectors (device)
  void *x = nullptr, *grad_f_x = nullptr, *y = nullptr, *z = nullptr, *res = nullptr, *z_prev = nullptr,
       *grad_f_z = nullptr;
  // scalars (held in double, always rounded through T)
  double gamma = 0, f_x = 0, g_z = 0, res_inf = 0, dot_gr = 0, res_sq = 0, beta = 0;
  double f_z = NAN, f_z_upp = NAN;
  int n_backtracks = 0, flags = 0;
  bool adaptive = false;
  bool initialized = false;
  // extrapolation sequence state (nesterov.jl)
  double seq_stepsize = -1, seq_theta = -1;  // AdaptiveNesterovSequence :56-60
  double seq_t = 1;                          // FixedNesterovSequence state
  int64_t seq_k = 1;                         // SimpleNesterovSequence state
  int64_t passes0 = 0;
  // residual reuse (adaptive FFB): A z - b and A z_prev - b, so that A x - b at the extrapolated point needs no pass
  void *rz = nullptr, *rz_prev = nullptr;
  bool rz_valid = false;
  bool defer_sync = false;  // pg_iter_run_batched: enqueue without reading the scalar block back
  // single-sweep iterations (pg_ls_fused_pass): the sweep of iteration k also produces the extrapolated point of
  // iteration k+1 and its residual.  sp_ready: that speculation is available -- fixed step: x_next, f->r = A x_next - b
  // (valid while f->r_gen == sp_gen), sp_f = f(x_next), the sequence state after its beta in spec_*; adaptive step:
  // rz = A z - b and sp_f = f(z).
  bool single_sweep = false;
  void* x_next = nullptr;
  bool sp_ready = false;
  uint64_t sp_gen = 0;
  double sp_f = 0, sp_beta = 0;
  int sp_slot = 0;
  int timeouts_in_a_row = 0;  // consecutive sweeps lost to a team timeout (three: the iterator stays with two sweeps)
  int fx_src = -1;  // deferred reads (pg_iter_run_batched): scalar slot that holds f(x) of the current state, -1 = f_x is current
  double spec_stepsize = -1, spec_theta = -1, spec_t = 1;
  int64_t spec_k = 1;
};


// ---- Nesterov sequences, evaluated in T like the reference's R (host and device) -------------------
template <typename T>
struct SeqState {
  T stepsize, theta, t;  // AdaptiveNesterovSequence (stepsize, theta) ; FixedNesterovSequence (t)
  long long k;           // SimpleNesterovSequence
};

template <typename T>
__host__ __device__ inline T seq_next_hd(int kind, T mf, T p0, T p1, SeqState<T>& st, T gamma, T host_beta) {
  switch (kind) {
    case PG_SEQ_FIXED: {  // nesterov.jl:14-17
      const T t = st.t;
      const T t_next = (T(1) + sqrt(T(1) + T(4) * t * t)) / T(2);
      st.t = t_next;
      return (t - T(1)) / t_next;
    }
    case PG_SEQ_SIMPLE: {  // nesterov.jl:36
      const long long k = st.k++;
      return (T)(k - 1) / (T)(k + 2);
    }
    case PG_SEQ_CONSTANT: {  // nesterov.jl:51-54
      const T k_inverse = p0 * p1;
      return (T(1) - sqrt(k_inverse)) / (T(1) + sqrt(k_inverse));
    }
    case PG_SEQ_HOST:
      return host_beta;
    case PG_SEQ_REPEATED:  // Iterators.repeated(beta)
      return p0;
    case PG_SEQ_ADAPTIVE:
    default: {  // nesterov.jl:89-103
      const T m = mf;
      const T stepsize = gamma;
      T s_step = st.stepsize, s_theta = st.theta;
      if (s_step < T(0)) {
        s_step = stepsize;
        s_theta = m > T(0) ? (T)sqrt(m * stepsize) : T(1);
      }
      const T b = s_theta * s_theta / s_step - m;
      const T delta = b * b + T(4) * (s_theta * s_theta) / (s_step * stepsize);
      const T theta = stepsize * (-b + sqrt(delta)) / T(2);
      const T beta = stepsize * s_theta * (T(1) - s_theta) / (s_step * theta + stepsize * s_theta * s_theta);
      st.stepsize = stepsize;
      st.theta = theta;
      return beta;
    }
  }
}


// the scalar block handed back by every pg_iter_* entry point
inline void fill_scalars(const pg_iter* it, pg_iter_scalars* s) {
  if (!s) return;
  s->gamma = it->gamma;
  s->f_x = it->f_x;
  s->g_z = it->g_z;
  s->res_inf = it->res_inf;
  s->beta = it->beta;
  s->f_z = it->f_z;
  s->f_z_upp = it->f_z_upp;
  s->n_backtracks = it->n_backtracks;
  s->flags = it->flags;
  s->a_passes = it->f->a_passes - it->passes0;
}
