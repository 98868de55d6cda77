// Fused ForwardBackward / FastForwardBackward iterations for f = LeastSquares, g in {Zero, NormL1, IndBox}.
//
// Host-side control flow restates the reference line by line; all array work is enqueued as the kernels of
// pg_gemv.hip / pg_vec.hip on the context stream.  Scalars that the reference holds in R = real(eltype(x0))
// (gamma, f_x, g_z, the Nesterov recurrences, the line-search compare) are held in T here as well.
//
//   init : forward_backward.jl:65-84   / fast_forward_backward.jl:73-97
//   step : forward_backward.jl:86-123  / fast_forward_backward.jl:106-145
//   line search : src/utilities/fb_tools.jl:3-5 (f_model), :7-12 (L estimate), :24-63 (backtrack_stepsize!)
//   sequences   : src/accel/nesterov.jl:14-17, :36, :51-54, :56-103
//   driver loop : src/ProximalAlgorithms.jl:114-123
#include <limits>
#include <utility>

#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "pg_iter_internal.h"

namespace {

template <typename T>
struct Arith {
  static double r(double v) { return (double)(T)v; }  // round to working precision
};

template <typename T>
double seq_next(pg_iter* it, double gamma_d, double host_beta) {
  SeqState<T> st{(T)it->seq_stepsize, (T)it->seq_theta, (T)it->seq_t, (long long)it->seq_k};
  const T beta = seq_next_hd<T>(it->o.seq_kind, (T)it->o.mf, (T)it->o.seq_p0, (T)it->o.seq_p1, st, (T)gamma_d, (T)host_beta);
  it->seq_stepsize = (double)st.stepsize;
  it->seq_theta = (double)st.theta;
  it->seq_t = (double)st.t;
  it->seq_k = st.k;
  return (double)beta;
}

inline size_t vec_bytes(const pg_iter* it) {
  return (size_t)pg_round_up((int64_t)((size_t)(it->n > 0 ? it->n : 1) * pg_sizeof(it->dtype)), 256);
}

// norm(res, Inf) as the stopping rule sees it (forward_backward.jl:126): a state whose <grad f(x), res> is NaN has not converged, whatever
// its res says.  After an overflow the one-read iteration can sit at x = z = 0 with a NaN gradient (its residual A x - b is carried
// along, not recomputed from x as in the reference: NaN stays), y = NaN, z = prox(NaN) = 0, res = 0 -- and "stop" (found by the
// option-drawing fuzzer of round 6 on runs whose minimum_gamma forces a diverging step).  NaN / gamma <= tol is false: such a run goes
// on to maxit like the reference's, which cycles through overflow and restart without ever meeting the rule.
static inline double res_inf_guarded(double res_inf, double dot_gr) { return dot_gr != dot_gr ? dot_gr : res_inf; }

// epilogue + read back {g_z, res_inf, <grad,res>, ||res||^2} and f (slot 0) in one copy
template <typename T>
pg_status epilogue_and_read(pg_iter* it, bool read_f) {
  pg_ctx* c = it->ctx;
  PG_TRY(pg_fb_epilogue_async(c, it->dtype, it->n, it->x, it->grad_f_x, it->gamma, it->o.g_kind, it->o.g_p0,
                              it->o.g_p1, it->y, it->z, it->res, it->g_v0, it->g_v1));
  PG_TRY(pg_ls_allreduce_epilogue_scalars(it->f));  // column shards: the kernel's sums cover this rank's columns only
  if (it->defer_sync) return PG_OK;  // scalars stay on the device side until the batch is synchronised
  PG_TRY(pg_read_scalars(c, PG_S_F, 5));
  if (read_f) it->f_x = Arith<T>::r(c->hscal[PG_S_F]);
  it->g_z = Arith<T>::r(c->hscal[PG_S_GZ]);
  it->res_inf = res_inf_guarded(Arith<T>::r(c->hscal[PG_S_RESINF]), c->hscal[PG_S_DOT]);
  it->dot_gr = Arith<T>::r(c->hscal[PG_S_DOT]);
  it->res_sq = Arith<T>::r(c->hscal[PG_S_RESSQ]);
  return PG_OK;
}

// f_model: fb_tools.jl:3-5   f_x - real(dot(grad, res)) + (L / 2) * norm(res)^2, L = alpha / gamma, alpha = 1
template <typename T>
double f_model(const pg_iter* it) {
  const T L = T(1) / (T)it->gamma;
  return (double)((T)it->f_x - (T)it->dot_gr + (L / T(2)) * (T)it->res_sq);
}

// backtrack_stepsize!: fb_tools.jl:24-63 with A === nothing.  On entry the epilogue scalars describe the
// current (x, grad_f_x, gamma_prev) triple; gamma has already been multiplied by increase_gamma by the
// caller (forward_backward.jl:91), exactly like the reference, where y/z/res are NOT recomputed for the
// increased gamma before the first test.  keep_grad: also produce grad f(z) into grad_f_z (FB).
template <typename T>
pg_status backtrack(pg_iter* it, bool keep_grad) {
  pg_ctx* c = it->ctx;
  const T eps = std::numeric_limits<T>::epsilon();
  const T min_gamma = (T)it->o.minimum_gamma, reduce = (T)it->o.reduce_gamma;
  T f_upp = (T)f_model<T>(it);  // :42
  // :44  f(z) (and grad f(z) when it is kept)
  if (keep_grad)
    PG_TRY(pg_ls_vg_async(it->f, it->z, it->grad_f_z));
  else
    PG_TRY(pg_ls_value_async(it->f, it->z));
  PG_TRY(pg_read_scalars(c, PG_S_F, 1));
  T f_z = (T)c->hscal[PG_S_F];
  T tol = T(10) * eps * (T(1) + std::fabs(f_z));  // :45
  it->n_backtracks = 0;
  while (f_z > f_upp + tol && (T)it->gamma >= min_gamma) {  // :46
    it->gamma = (double)((T)it->gamma * reduce);            // :47
    PG_TRY(epilogue_and_read<T>(it, false));                // :48-50 y, z, g_z, res
    f_upp = (T)f_model<T>(it);                              // :51
    if (keep_grad)
      PG_TRY(pg_ls_vg_async(it->f, it->z, it->grad_f_z));   // :53
    else
      PG_TRY(pg_ls_value_async(it->f, it->z));
    PG_TRY(pg_read_scalars(c, PG_S_F, 1));
    f_z = (T)c->hscal[PG_S_F];
    tol = T(10) * eps * (T(1) + std::fabs(f_z));  // :54
    it->n_backtracks++;
  }
  if (!keep_grad && it->rz != nullptr && it->f->A->m > 0) {  // residual A z - b of the accepted trial
    PG_HIP(hipMemcpyAsync(it->rz, it->f->r, (size_t)it->f->A->m * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
    it->rz_valid = true;
  }
  if ((T)it->gamma < min_gamma) it->flags |= PG_FLAG_GAMMA_TOO_SMALL;  // :59-61 (@warn)
  it->f_z = (double)f_z;
  it->f_z_upp = (double)f_upp;
  return PG_OK;
}

template <typename T>
pg_status iter_init(pg_iter* it, const void* x0) {
  pg_ctx* c = it->ctx;
  const int64_t n = it->n;
  const size_t nb = (size_t)n * sizeof(T);
  it->flags = 0;
  it->n_backtracks = 0;
  it->beta = 0;
  it->f_z = it->f_z_upp = NAN;
  it->seq_stepsize = it->seq_theta = -1;
  it->seq_t = 1;
  it->seq_k = 1;
  it->passes0 = it->f->a_passes;
  it->sp_ready = false;
  it->sp_slot = 0;
  // x = copy(x0)                                                         fb:66 / ffb:74
  if (n > 0) PG_HIP(hipMemcpyAsync(it->x, x0, nb, hipMemcpyDeviceToDevice, c->stream));
  // f_x, grad_f_x = value_and_gradient(f, x)                             fb:67 / ffb:75
  PG_TRY(pg_ls_vg_async(it->f, it->x, it->grad_f_x));
  PG_TRY(pg_read_scalars(c, PG_S_F, 1));
  it->f_x = Arith<T>::r(c->hscal[PG_S_F]);
  it->rz_valid = false;
  if (it->rz_prev != nullptr && it->f->A->m > 0) {  // z_prev = copy(x): its residual is the one just computed
    PG_HIP(hipMemcpyAsync(it->rz_prev, it->f->r, (size_t)it->f->A->m * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
  }
  // gamma = iter.gamma === nothing ? 1 / lower_bound_smoothness_constant(f, I, x, grad_f_x) : iter.gamma
  double gamma = it->o.gamma > 0 ? it->o.gamma : (it->o.Lf > 0 ? (double)(T(1) / (T)it->o.Lf) : -1.0);
  if (gamma <= 0) {
    // fb_tools.jl:7-12: xeps = x .+ 1 ; grad at xeps ; norm(grad_eps - grad) / sqrt(length(x))
    // scratch: y <- xeps, res <- grad_eps, z <- grad_eps - grad   (all overwritten by the epilogue below)
    PG_TRY(pg_add_scalar(c, it->dtype, n, it->y, it->x, 1.0));
    PG_TRY(pg_ls_vg_async(it->f, it->y, it->res));
    PG_TRY(pg_axpby(c, it->dtype, n, it->z, 1.0, it->res, -1.0, it->grad_f_x));
    double nrm2 = 0;
    PG_TRY(pg_nrm2sq(c, it->dtype, n, it->z, &nrm2));
    double n_glob = (double)n;
    if (pg_col_sharded(c)) {  // the n-vector is distributed: sum the squared norms and the lengths of the slices
      c->hscal[PG_S_GZ] = c->hscal[PG_S_RESINF] = 0.0;
      c->hscal[PG_S_DOT] = (double)n;  // mapped host memory: visible to the kernels launched next
      c->hscal[PG_S_RESSQ] = nrm2;
      PG_TRY(pg_ls_allreduce_epilogue_scalars(it->f));
      PG_TRY(pg_read_scalars(c, PG_S_DOT, 2));
      n_glob = c->hscal[PG_S_DOT];
      nrm2 = c->hscal[PG_S_RESSQ];
    }
    const T Lest = (T)std::sqrt((T)nrm2) / (T)std::sqrt(n_glob);
    gamma = (double)(T(1) / Lest);
  }
  it->gamma = Arith<T>::r(gamma);
  // y = x - gamma .* grad_f_x ; z, g_z = prox(g, y, gamma) ; res = x - z    fb:71-72,81 / ffb:79-80,89
  PG_TRY(epilogue_and_read<T>(it, false));
  if (it->o.fast && n > 0)  // z_prev = copy(x)                            ffb:69
    PG_HIP(hipMemcpyAsync(it->z_prev, it->x, nb, hipMemcpyDeviceToDevice, c->stream));
  it->initialized = true;
  return PG_OK;
}

// ---------------------------------------------------------------------------------------------
// Single-sweep iterations: ONE read of A per iteration (pg_ls_fused_pass_async).  The sweep that forms A' r for the
// current point also runs the epilogue column by column and accumulates the residual of the NEXT point from the
// column while it is still in registers.  What the reference does in the order
//     [extrapolate :135 | A x :138] [A' r :138-139 | prox :140-142]          (one iteration)
// is executed as            ... A x ] [A' r | prox | extrapolate | A x_next] [ ...      (one sweep)
// i.e. a sweep is the second half of iteration k and the first half of iteration k+1.  The first half is speculative:
// if iteration k satisfies the stop rule it is simply never committed (x_next is a separate buffer, the sequence is
// advanced on a copy), so the state returned is exactly the reference's state k.
// ---------------------------------------------------------------------------------------------
template <typename T>
SeqState<T> seq_load(const pg_iter* it) {
  return SeqState<T>{(T)it->seq_stepsize, (T)it->seq_theta, (T)it->seq_t, (long long)it->seq_k};
}

template <typename T>
pg_status read_sweep_scalars(pg_iter* it) {
  pg_ctx* c = it->ctx;
  PG_TRY(pg_read_scalars(c, PG_S_F, PG_S_COUNT));
  it->g_z = Arith<T>::r(c->hscal[PG_S_GZ]);
  it->res_inf = res_inf_guarded(Arith<T>::r(c->hscal[PG_S_RESINF]), c->hscal[PG_S_DOT]);
  it->dot_gr = Arith<T>::r(c->hscal[PG_S_DOT]);
  it->res_sq = Arith<T>::r(c->hscal[PG_S_RESSQ]);
  it->sp_f = Arith<T>::r(c->hscal[PG_S_FNEXT + it->sp_slot]);
  return PG_OK;
}

// The sweep of this iteration could not be used: refused at launch (PG_ERR_UNSUPPORTED: nothing was written; it would be
// refused again, so the iterator leaves the single-sweep mode) or one of its workgroup teams timed out (PG_ERR_TIMEOUT:
// grad, y, z, res, x_next and the sweep's residual output are garbage; x, z_prev and, when the sweep wrote its residual
// elsewhere, f->r are intact).  Nothing of the sweep had been committed -- its outputs only become state when the NEXT
// step swaps them in -- so the second half of the iteration (fast_forward_backward.jl:138-142 / forward_backward.jl:113-120)
// is redone here the reference's way, A x then A' r, and the step reports PG_FLAG_SWEEP_FALLBACK.  With column shards every
// rank gets here in the same step (the timeout flag travels with the all-reduce payload), so the collectives still pair up.
template <typename T>
pg_status redo_with_two_sweeps(pg_iter* it, pg_status why, bool residual_intact) {
  pg_ctx* c = it->ctx;
  c->team_timeout = false;
  it->sp_ready = false;
  it->flags |= PG_FLAG_SWEEP_FALLBACK;
  // three lost sweeps in a row are not bad luck (a device shared with another process: its kernels and ours alternate, so
  // the members of a team never run together): stay with two sweeps instead of paying the bounded wait in every step.
  // The count is the same on every rank of a sharded job (the flag is exchanged), so they leave the mode together.
  if (why == PG_ERR_TIMEOUT && ++it->timeouts_in_a_row >= 3) it->single_sweep = false;
  if (why == PG_ERR_UNSUPPORTED) {
    // refused: nothing of the sweep ran, and it would be refused again -- also inside a batch (defer_sync): the two sweeps
    // are enqueued in its place and the batch's one read-back takes f(x) from PG_S_F like any two-sweep iteration.  With
    // column shards the refusal of ONE rank reaches every rank through the payload's flag (pg_gemv.hip), as
    // PG_ERR_UNSUPPORTED at the scalar read-back, i.e. here -- outside a batch only; inside one the batch fails with that
    // code on every rank and the caller restarts it step by step (algorithm.py).
    it->single_sweep = false;
    it->fx_src = -1;
  }
  if (residual_intact) {  // f->r = A x - b and dscal[PG_S_F] = f(x) are still those of x: only A' r is missing
    PG_TRY(pg_ls_grad_stage_async(it->f, it->grad_f_x));
  } else {
    PG_TRY(pg_ls_vg_async(it->f, it->x, it->grad_f_x));
  }
  if (it->rz_prev != nullptr) it->rz_valid = false;
  return epilogue_and_read<T>(it, true);
}

static inline bool sweep_lost(pg_status st) { return st == PG_ERR_UNSUPPORTED || st == PG_ERR_TIMEOUT; }

template <typename T>
pg_status iter_step_single_sweep(pg_iter* it) {
  pg_ctx* c = it->ctx;
  pg_ls* f = it->f;
  const pg_iter_opts& o = it->o;
  if (!it->adaptive) {
    const double gfix = (o.gamma > 0 || o.Lf > 0) ? Arith<T>::r(o.gamma > 0 ? o.gamma : (double)(T(1) / (T)o.Lf)) : it->gamma;
    if (o.fast) {
      // ---- FastForwardBackward, fixed step: fast_forward_backward.jl:131-142 ----
      const bool fresh_first_half = !(it->sp_ready && it->sp_gen == f->r_gen);
      if (fresh_first_half) {  // first half of this iteration (:131-138) on its own
        SeqState<T> s = seq_load<T>(it);
        it->sp_beta = (double)seq_next_hd<T>(o.seq_kind, (T)o.mf, (T)o.seq_p0, (T)o.seq_p1, s, (T)gfix, T(0));  // :134
        it->spec_stepsize = (double)s.stepsize, it->spec_theta = (double)s.theta, it->spec_t = (double)s.t, it->spec_k = s.k;
        PG_TRY(pg_extrapolate(c, it->dtype, it->n, it->x_next, it->z, it->z_prev, it->sp_beta));  // :135
        PG_TRY(pg_ls_value_async(f, it->x_next));                                                 // :138 (A x - b, f)
        PG_TRY(pg_read_scalars(c, PG_S_F, 1));
        it->sp_f = Arith<T>::r(c->hscal[PG_S_F]);
      }
      // commit the first half
      it->gamma = gfix;                                                                           // :131
      it->beta = it->sp_beta;
      it->seq_stepsize = it->spec_stepsize, it->seq_theta = it->spec_theta, it->seq_t = it->spec_t, it->seq_k = it->spec_k;
      std::swap(it->x, it->x_next);
      std::swap(it->z_prev, it->z);                                                               // :136
      if (fresh_first_half || !it->defer_sync) {
        it->f_x = it->sp_f;
        it->fx_src = -1;
      } else {
        it->fx_src = PG_S_FNEXT + it->sp_slot;  // written by the previous sweep, read back with the batch
      }
      // second half (:138-142) + the next iteration's first half, one sweep
      SeqState<T> s2 = seq_load<T>(it);
      const double beta2 = (double)seq_next_hd<T>(o.seq_kind, (T)o.mf, (T)o.seq_p0, (T)o.seq_p1, s2, (T)it->gamma, T(0));
      it->sp_slot ^= 1;
      pg_status st = pg_ls_fused_pass_async(f, nullptr, nullptr, c->dscal + PG_S_FNEXT + it->sp_slot, it->x, it->z_prev,
                                            it->gamma, beta2, o.g_kind, o.g_p0, o.g_p1, it->grad_f_x, it->y, it->z, it->res,
                                            it->x_next, it->g_v0, it->g_v1);
      if (st == PG_OK && !it->defer_sync) st = read_sweep_scalars<T>(it);
      if (sweep_lost(st) && (!it->defer_sync || st == PG_ERR_UNSUPPORTED)) return redo_with_two_sweeps<T>(it, st, false);
      PG_TRY(st);
      it->sp_beta = beta2;
      it->spec_stepsize = (double)s2.stepsize, it->spec_theta = (double)s2.theta, it->spec_t = (double)s2.t, it->spec_k = s2.k;
      it->sp_gen = f->r_gen;
      it->sp_ready = true;
      it->timeouts_in_a_row = 0;
    } else {
      // ---- ForwardBackward, fixed step: forward_backward.jl:111-120 (the next point is the prox output itself) ----
      const bool fresh_first_half = !(it->sp_ready && it->sp_gen == f->r_gen);
      if (fresh_first_half) {
        PG_TRY(pg_ls_value_async(f, it->z));
        PG_TRY(pg_read_scalars(c, PG_S_F, 1));
        it->sp_f = Arith<T>::r(c->hscal[PG_S_F]);
      }
      std::swap(it->x, it->z);  // :112
      if (fresh_first_half || !it->defer_sync) {  // :113
        it->f_x = it->sp_f;
        it->fx_src = -1;
      } else {
        it->fx_src = PG_S_FNEXT + it->sp_slot;
      }
      it->sp_slot ^= 1;
      pg_status st = pg_ls_fused_pass_async(f, nullptr, nullptr, c->dscal + PG_S_FNEXT + it->sp_slot, it->x, it->x, it->gamma,
                                            0.0, o.g_kind, o.g_p0, o.g_p1, it->grad_f_x, it->y, it->z, it->res, nullptr, it->g_v0,
                                            it->g_v1);
      if (st == PG_OK && !it->defer_sync) st = read_sweep_scalars<T>(it);
      if (sweep_lost(st) && (!it->defer_sync || st == PG_ERR_UNSUPPORTED)) return redo_with_two_sweeps<T>(it, st, false);
      PG_TRY(st);
      it->sp_gen = f->r_gen;
      it->sp_ready = true;
      it->timeouts_in_a_row = 0;
    }
    return PG_OK;
  }
  // ---- FastForwardBackward, adaptive step with the residual pair: fast_forward_backward.jl:110-142 ----
  const T eps = std::numeric_limits<T>::epsilon();
  const T min_gamma = (T)o.minimum_gamma, reduce = (T)o.reduce_gamma;
  const size_t mb = (size_t)f->A->m * sizeof(T);
  it->gamma = (double)((T)it->gamma * (T)o.increase_gamma);  // :111
  T f_upp = (T)f_model<T>(it);                               // fb_tools.jl:42
  T f_z;
  if (it->sp_ready) {
    f_z = (T)it->sp_f;  // the last sweep left A z - b in rz and f(z) with it: no pass for the first trial
  } else {
    PG_TRY(pg_ls_value_async(f, it->z));
    if (mb) PG_HIP(hipMemcpyAsync(it->rz, f->r, mb, hipMemcpyDeviceToDevice, c->stream));
    PG_TRY(pg_read_scalars(c, PG_S_F, 1));
    f_z = (T)c->hscal[PG_S_F];
  }
  T tol = T(10) * eps * (T(1) + std::fabs(f_z));
  while (f_z > f_upp + tol && (T)it->gamma >= min_gamma) {  // fb_tools.jl:46-55
    it->gamma = (double)((T)it->gamma * reduce);
    PG_TRY(epilogue_and_read<T>(it, false));
    f_upp = (T)f_model<T>(it);
    PG_TRY(pg_ls_value_async(f, it->z));
    if (mb) PG_HIP(hipMemcpyAsync(it->rz, f->r, mb, hipMemcpyDeviceToDevice, c->stream));
    PG_TRY(pg_read_scalars(c, PG_S_F, 1));
    f_z = (T)c->hscal[PG_S_F];
    tol = T(10) * eps * (T(1) + std::fabs(f_z));
    it->n_backtracks++;
  }
  if ((T)it->gamma < min_gamma) it->flags |= PG_FLAG_GAMMA_TOO_SMALL;
  it->f_z = (double)f_z;
  it->f_z_upp = (double)f_upp;
  it->beta = seq_next<T>(it, it->gamma, 0.0);                                                       // :134
  PG_TRY(pg_extrapolate(c, it->dtype, it->n, it->x, it->z, it->z_prev, it->beta));                  // :135
  std::swap(it->z_prev, it->z);                                                                     // :136
  // A x - b = (1 + beta)(A z - b) - beta (A z_prev - b)   (:138 without reading A)
  // (row teams: the combination covers this device's rows; f = the sum over the devices, exchanged like the sweep's own f)
  const bool rteam = pg_row_sharded(c) && pg_rteam_active(c);
  PG_TRY(pg_residual_combo_async(c, it->dtype, f->A->m, f->r, (double)(T(1) + (T)it->beta), it->rz,
                                 (double)(-(T)it->beta), it->rz_prev, 0.5 * f->lam, nullptr, rteam ? c->rteam.f_local : nullptr));
  if (rteam) PG_TRY(pg_rteam_sum_scalar(c, c->rteam.f_local, c->dscal + PG_S_F));
  f->r_gen++;
  std::swap(it->rz_prev, it->rz);  // the residual at the new z_prev
  // A' r, prox (:138-142) and the residual of the NEW z for the next line search, one sweep
  it->sp_slot ^= 1;
  pg_status st = pg_ls_fused_pass_async(f, f->r, it->rz, c->dscal + PG_S_FNEXT + it->sp_slot, it->x, it->z_prev, it->gamma, 0.0,
                                        o.g_kind, o.g_p0, o.g_p1, it->grad_f_x, it->y, it->z, it->res, nullptr, it->g_v0, it->g_v1);
  if (st == PG_OK) st = read_sweep_scalars<T>(it);
  if (sweep_lost(st)) {  // the sweep read f->r and wrote rz: the residual of x and f(x) (PG_S_F) are intact
    it->rz_valid = false;
    return redo_with_two_sweeps<T>(it, st, true);
  }
  PG_TRY(st);
  it->sp_ready = true;
  it->timeouts_in_a_row = 0;
  it->rz_valid = false;
  it->f_x = Arith<T>::r(c->hscal[PG_S_F]);  // from the residual combination
  return PG_OK;
}

template <typename T>
pg_status iter_step(pg_iter* it, double host_beta) {
  it->flags = 0;
  if (it->ctx->coop_slow) {  // (pg_gemv_tn2.hip::launch_tnt switched this context to plain launches: say so once)
    it->flags |= PG_FLAG_COOP_SLOW;
    it->ctx->coop_slow = false;
  }
  it->n_backtracks = 0;
  it->f_z = it->f_z_upp = NAN;
  if (it->single_sweep && !(it->defer_sync && it->adaptive)) return iter_step_single_sweep<T>(it);
  it->sp_ready = false;
  if (!it->o.fast) {
    // ---------------- ForwardBackward: forward_backward.jl:86-123 ----------------
    if (it->adaptive) {
      it->gamma = (double)((T)it->gamma * (T)it->o.increase_gamma);  // :91
      PG_TRY(backtrack<T>(it, true));                                // :92-108
      it->f_x = it->f_z;                                             // :92 (state.f_x = f_Az)
      std::swap(it->x, it->z);                                       // :109
      std::swap(it->grad_f_x, it->grad_f_z);                         // :110
      PG_TRY(epilogue_and_read<T>(it, false));                       // :117-120
    } else {
      std::swap(it->x, it->z);                                       // :112
      PG_TRY(pg_ls_vg_async(it->f, it->x, it->grad_f_x));            // :113-114
      PG_TRY(epilogue_and_read<T>(it, true));                        // :117-120
    }
  } else {
    // ---------------- FastForwardBackward: fast_forward_backward.jl:106-145 ----------------
    if (it->adaptive) {
      it->gamma = (double)((T)it->gamma * (T)it->o.increase_gamma);  // :111
      PG_TRY(backtrack<T>(it, false));                               // :112-127 (grad at z discarded)
    } else {
      if (it->o.gamma > 0 || it->o.Lf > 0)                           // :131 (state.gamma = iter.gamma)
        it->gamma = Arith<T>::r(it->o.gamma > 0 ? it->o.gamma : (double)(T(1) / (T)it->o.Lf));
    }
    it->beta = seq_next<T>(it, it->gamma, host_beta);                // :134
    PG_TRY(pg_extrapolate(it->ctx, it->dtype, it->n, it->x, it->z, it->z_prev, it->beta));  // :135
    std::swap(it->z_prev, it->z);                                    // :136
    if (it->adaptive && it->rz != nullptr && it->rz_valid) {
      // :138-139 without re-reading A for A*x:  A x - b = (1 + beta)(A z - b) - beta (A z_prev - b); the line
      // search has just produced A z - b.  One pass (A' r) instead of two.
      pg_ls* f = it->f;
      void* f_typed = pg_row_sharded(it->ctx) ? (void*)((char*)f->gbuf + (size_t)f->A->n * sizeof(T)) : nullptr;
      PG_TRY(pg_residual_combo_async(it->ctx, it->dtype, f->A->m, f->r, (double)(T(1) + (T)it->beta),
                                     it->rz, (double)(-(T)it->beta), it->rz_prev, 0.5 * f->lam, f_typed));
      PG_TRY(pg_ls_grad_stage_async(f, it->grad_f_x));
      std::swap(it->rz_prev, it->rz);  // the residual at the new z_prev
      it->rz_valid = false;
    } else {
      PG_TRY(pg_ls_vg_async(it->f, it->x, it->grad_f_x));            // :138-139
      if (it->rz_prev != nullptr) it->rz_valid = false;
    }
    PG_TRY(epilogue_and_read<T>(it, true));                          // :140-142
  }
  return PG_OK;
}

}  // namespace

extern "C" {

pg_status pg_iter_opts_default(pg_iter_opts* o) {
  PG_REQUIRE(o != nullptr, "opts is null");
  memset(o, 0, sizeof(*o));
  o->fast = 0;
  o->adaptive = -1;
  o->Lf = -1;
  o->gamma = -1;
  o->minimum_gamma = 1e-7;   // forward_backward.jl:45
  o->reduce_gamma = 0.5;     // :46
  o->increase_gamma = 1.0;   // :47
  o->mf = 0;                 // fast_forward_backward.jl:48
  o->seq_kind = PG_SEQ_ADAPTIVE;
  o->g_kind = PG_G_ZERO;
  o->reuse_residual = 1;
  o->single_sweep = 1;
  return PG_OK;
}

pg_status pg_iter_create(pg_ctx* c, pg_ls* f, const pg_iter_opts* o, pg_iter** out) {
  PG_REQUIRE(c != nullptr && f != nullptr && o != nullptr && out != nullptr, "null argument");
  PG_REQUIRE(f->ctx == c, "f belongs to another context");
  PG_REQUIRE(o->g_kind == PG_G_ZERO || o->g_kind == PG_G_NORML1 || o->g_kind == PG_G_INDBOX, "unknown g_kind");
  PG_REQUIRE(o->seq_kind >= PG_SEQ_ADAPTIVE && o->seq_kind <= PG_SEQ_REPEATED, "unknown seq_kind");
  *out = nullptr;
  pg_iter* it = new pg_iter();
  it->ctx = c;
  it->f = f;
  it->o = *o;
  it->dtype = f->A->dtype;
  it->n = f->A->n;
  // adaptive::Bool = gamma === nothing        forward_backward.jl:43-44
  const bool gamma_known = (o->gamma > 0) || (o->Lf > 0);
  it->adaptive = o->adaptive < 0 ? !gamma_known : (o->adaptive != 0);
  if (pg_col_sharded(c) && it->adaptive && !(o->fast && o->reuse_residual != 0)) {
    delete it;
    pg_set_error("column-sharded operators support the adaptive step for FastForwardBackward with reuse_residual only");
    return PG_ERR_UNSUPPORTED;
  }
  // experiment hooks: PG_ITER_VEC_SKEW = extra bytes between consecutive state vectors, PG_ITER_BASE_SKEW = offset of the
  // first one within the allocation (both rounded to 256 B)
  auto env_bytes = [](const char* name) -> size_t {
    const char* v = env_str(name);
    return v ? (size_t)pg_round_up((int64_t)strtoll(v, nullptr, 10), 256) : 0;
  };
  const size_t base_skew = env_bytes("PG_ITER_BASE_SKEW");
  const size_t vb = vec_bytes(it) + env_bytes("PG_ITER_VEC_SKEW");
  const bool reuse = o->fast && o->reuse_residual != 0;
  // one read of A per iteration where the fused sweep applies: FB / FFB with a fixed step, FFB adaptive with the residual
  // pair; host-provided extrapolation coefficients arrive one step at a time, so they need the two-sweep path
  if (o->single_sweep != 0 && pg_row_sharded(c) && pg_rteam_active(c) && f->A->m > 0 && f->A->n > 0) {
    // row teams: the devices agree on the longest block of the team before anyone decides how it sweeps -- a collective (every
    // device creates its iterator at the same point of the program); its failure is this call's failure, on this rank, now
    const pg_status agreed = pg_mat_row_team_agree(c, f->A);
    if (agreed != PG_OK) {
      delete it;
      return agreed;
    }
  }
  it->single_sweep = o->single_sweep != 0 && pg_ls_fused_pass_supported(f) && !(o->fast && o->seq_kind == PG_SEQ_HOST) &&
                     (!it->adaptive || (o->fast && reuse));
  const int nvec = it->single_sweep ? 7 : 6;
  const size_t mb = reuse ? (size_t)pg_round_up((int64_t)((size_t)(f->A->m > 0 ? f->A->m : 1) * pg_sizeof(it->dtype)), 256) : 0;
  PG_HIP(hipSetDevice(c->device));
  hipError_t e = hipMalloc(&it->slab, base_skew + vb * nvec + 2 * mb);
  if (e != hipSuccess) {
    pg_set_error("state allocation (%zu bytes) failed: %s", vb * nvec, hipGetErrorString(e));
    delete it;
    return PG_ERR_ALLOC;
  }
  e = hipMemsetAsync(it->slab, 0, base_skew + vb * nvec + 2 * mb, c->stream);
  if (e != hipSuccess) {
    (void)hipFree(it->slab);
    delete it;
    pg_set_error("hipMemsetAsync failed: %s", hipGetErrorString(e));
    return PG_ERR_HIP;
  }
  char* base = (char*)it->slab + base_skew;
  it->x = base + 0 * vb;
  it->grad_f_x = base + 1 * vb;
  it->y = base + 2 * vb;
  it->z = base + 3 * vb;
  it->res = base + 4 * vb;
  if (o->fast)
    it->z_prev = base + 5 * vb;
  else
    it->grad_f_z = base + 5 * vb;
  if (it->single_sweep) it->x_next = base + 6 * vb;
  if (reuse) {
    it->rz = base + nvec * vb;
    it->rz_prev = base + nvec * vb + mb;
  }
  *out = it;
  return PG_OK;
}

pg_status pg_iter_set_g_vectors(pg_iter* it, const void* lo, const void* hi) {
  PG_REQUIRE(it != nullptr, "iterator is null");
  if (it->o.g_kind == PG_G_NORML1) {  // per-element weights lam_j in the first vector
    PG_REQUIRE(hi == nullptr, "g = NormL1 takes one vector (the weights); the second must be null");
  } else {
    PG_REQUIRE((lo == nullptr) == (hi == nullptr), "lo and hi must both be vectors or both be null");
    PG_REQUIRE(lo == nullptr || it->o.g_kind == PG_G_INDBOX, "per-element parameters are for g = IndBox or NormL1");
  }
  it->g_v0 = lo;
  it->g_v1 = hi;
  return PG_OK;
}

pg_status pg_iter_destroy(pg_iter* it) {
  if (!it) return PG_OK;
  if (it->slab) {
    (void)hipStreamSynchronize(it->ctx->stream);
    (void)hipFree(it->slab);
  }
  delete it;
  return PG_OK;
}

pg_status pg_iter_init(pg_iter* it, const void* x0, pg_iter_scalars* out) {
  PG_REQUIRE(it != nullptr, "iterator is null");
  PG_REQUIRE(it->n == 0 || x0 != nullptr, "x0 is null");
  PG_TRY(it->dtype == PG_F32 ? iter_init<float>(it, x0) : iter_init<double>(it, x0));
  fill_scalars(it, out);
  return PG_OK;
}

pg_status pg_iter_step(pg_iter* it, double host_beta, pg_iter_scalars* out) {
  PG_REQUIRE(it != nullptr, "iterator is null");
  PG_REQUIRE(it->initialized, "pg_iter_init has not been called");
  PG_TRY(it->dtype == PG_F32 ? iter_step<float>(it, host_beta) : iter_step<double>(it, host_beta));
  fill_scalars(it, out);
  return PG_OK;
}

// IterativeAlgorithm: for (k, state) in enumerate(iter): if k >= maxit || stop(iter, state) return (.., k)
pg_status pg_iter_run(pg_iter* it, int64_t k_start, int64_t maxit, double tol, int64_t* k_out,
                      pg_iter_scalars* out) {
  PG_REQUIRE(it != nullptr && k_out != nullptr, "null argument");
  PG_REQUIRE(it->initialized, "pg_iter_init has not been called");
  PG_REQUIRE(it->o.seq_kind != PG_SEQ_HOST || !it->o.fast, "PG_SEQ_HOST needs per-step coefficients");
  int64_t k = k_start;
  const bool f32 = it->dtype == PG_F32;
  for (;;) {
    // default_stopping_criterion: norm(res, Inf) / gamma <= tol   (in R)
    const bool stop = f32 ? ((float)it->res_inf / (float)it->gamma <= (float)tol)
                          : (it->res_inf / it->gamma <= tol);
    if (k >= maxit || stop) break;
    PG_TRY(f32 ? iter_step<float>(it, 0.0) : iter_step<double>(it, 0.0));
    ++k;
  }
  *k_out = k;
  fill_scalars(it, out);
  return PG_OK;
}

// Fixed-step iterations need no scalar from the previous one (gamma is constant, beta comes from gamma), so a batch
// of `check_every` iterations is enqueued back to back and the host synchronises once per batch: the stopping rule
// is then evaluated every `check_every` iterations (k_out is a multiple of the batch size past k_start, or maxit).
pg_status pg_iter_run_batched(pg_iter* it, int64_t k_start, int64_t maxit, double tol, int32_t check_every,
                              int64_t* k_out, pg_iter_scalars* out) {
  PG_REQUIRE(it != nullptr && k_out != nullptr, "null argument");
  PG_REQUIRE(it->initialized, "pg_iter_init has not been called");
  PG_REQUIRE(check_every >= 1, "check_every must be >= 1");
  PG_REQUIRE(!it->adaptive, "batched runs need a fixed step (the line search is a host decision per iteration)");
  PG_REQUIRE(it->o.seq_kind != PG_SEQ_HOST || !it->o.fast, "PG_SEQ_HOST needs per-step coefficients");
  int64_t k = k_start;
  const bool f32 = it->dtype == PG_F32;
  pg_ctx* c = it->ctx;
  for (;;) {
    const bool stop = f32 ? ((float)it->res_inf / (float)it->gamma <= (float)tol)
                          : (it->res_inf / it->gamma <= tol);
    if (k >= maxit || stop) break;
    int64_t nb = maxit - k < check_every ? maxit - k : check_every;
    it->defer_sync = true;
    pg_status st = PG_OK;
    for (int64_t j = 0; j < nb && st == PG_OK; ++j) st = f32 ? iter_step<float>(it, 0.0) : iter_step<double>(it, 0.0);
    it->defer_sync = false;
    PG_TRY(st);
    k += nb;
    // one synchronisation per batch: the scalar block now describes the last enqueued iteration
    PG_TRY(pg_read_scalars(c, PG_S_F, PG_S_COUNT));
    if (it->single_sweep) {  // f(x) of the last state came from the sweep before the last one; the last sweep's is speculative
      if (it->fx_src >= 0) it->f_x = f32 ? (double)(float)c->hscal[it->fx_src] : c->hscal[it->fx_src];
      it->fx_src = -1;
      it->sp_f = f32 ? (double)(float)c->hscal[PG_S_FNEXT + it->sp_slot] : c->hscal[PG_S_FNEXT + it->sp_slot];
    } else
      it->f_x = f32 ? (double)(float)c->hscal[PG_S_F] : c->hscal[PG_S_F];
    it->g_z = f32 ? (double)(float)c->hscal[PG_S_GZ] : c->hscal[PG_S_GZ];
    it->res_inf = res_inf_guarded(f32 ? (double)(float)c->hscal[PG_S_RESINF] : c->hscal[PG_S_RESINF], c->hscal[PG_S_DOT]);
    it->dot_gr = f32 ? (double)(float)c->hscal[PG_S_DOT] : c->hscal[PG_S_DOT];
    it->res_sq = f32 ? (double)(float)c->hscal[PG_S_RESSQ] : c->hscal[PG_S_RESSQ];
  }
  *k_out = k;
  fill_scalars(it, out);
  return PG_OK;
}

// ---------------------------------------------------------------------------------------------
// Checkpoint / resume.  The reference keeps ALL algorithm memory in the state struct (forward_backward.jl:52-63,
// fast_forward_backward.jl:60-71 incl. the mutable AdaptiveNesterovSequence, nesterov.jl:56-60), so `iterate(iter, saved)`
// resumes a solve.  Here that memory is: the state vectors, the residual vectors (A x - b; for the adaptive fast iteration
// the pair A z - b, A z_prev - b), the scalars, the sequence state and -- single-sweep iterations -- the speculative first
// half of the next iteration (x_next, its residual, the sequence state after its coefficient), which the last sweep produced
// with its own summation order: dropping it would resume correctly but not bit-identically.  One host blob carries it all.
// ---------------------------------------------------------------------------------------------
namespace {
struct StateBlobHeader {
  uint32_t magic, version;
  int32_t dtype, fast, adaptive, single_sweep, reuse, g_kind;
  int64_t n, m, ld, bytes;
  double gamma, f_x, g_z, res_inf, dot_gr, res_sq, beta, f_z, f_z_upp;
  double seq_stepsize, seq_theta, seq_t;
  int64_t seq_k;
  double sp_f, sp_beta, spec_stepsize, spec_theta, spec_t;
  int64_t spec_k, a_passes;
  int32_t flags, n_backtracks, rz_valid, sp_ready, sp_slot, r_current, reserved0, reserved1;
};
constexpr uint32_t STATE_MAGIC = 0x54534750u;  // "PGST"

inline int64_t state_blob_bytes(const pg_iter* it) {
  const size_t s = pg_sizeof(it->dtype);
  const int nvec = 6 + (it->x_next != nullptr ? 1 : 0);
  const pg_mat* A = it->f->A;
  return (int64_t)(sizeof(StateBlobHeader) + (size_t)nvec * (size_t)it->n * s + (size_t)A->ld * s +
                   (it->rz != nullptr ? 2 * (size_t)A->m * s : 0));
}

// the vectors of the blob in their fixed order
inline int state_vectors(pg_iter* it, void** ptrs, size_t* bytes) {
  const size_t s = pg_sizeof(it->dtype);
  const size_t nb = (size_t)it->n * s;
  int k = 0;
  auto add = [&](void* p, size_t b) { ptrs[k] = p, bytes[k] = b, ++k; };
  add(it->x, nb), add(it->grad_f_x, nb), add(it->y, nb), add(it->z, nb), add(it->res, nb);
  add(it->o.fast ? it->z_prev : it->grad_f_z, nb);
  if (it->x_next != nullptr) add(it->x_next, nb);
  add(it->f->r, (size_t)it->f->A->ld * s);
  if (it->rz != nullptr) add(it->rz, (size_t)it->f->A->m * s), add(it->rz_prev, (size_t)it->f->A->m * s);
  return k;
}
}  // namespace

pg_status pg_iter_state_bytes(pg_iter* it, int64_t* bytes_out) {
  PG_REQUIRE(it != nullptr && bytes_out != nullptr, "null argument");
  *bytes_out = state_blob_bytes(it);
  return PG_OK;
}

pg_status pg_iter_state_download(pg_iter* it, void* host_blob, int64_t bytes) {
  PG_REQUIRE(it != nullptr && host_blob != nullptr, "null argument");
  PG_REQUIRE(it->initialized, "pg_iter_init has not been called");
  PG_REQUIRE(bytes >= state_blob_bytes(it), "the blob is smaller than pg_iter_state_bytes");
  PG_REQUIRE(it->o.seq_kind != PG_SEQ_HOST || !it->o.fast, "a host-drawn extrapolation sequence lives in the caller: nothing to save here");
  pg_ctx* c = it->ctx;
  StateBlobHeader h;
  memset(&h, 0, sizeof(h));
  h.magic = STATE_MAGIC, h.version = 1;
  h.dtype = it->dtype, h.fast = it->o.fast, h.adaptive = it->adaptive ? 1 : 0, h.single_sweep = it->single_sweep ? 1 : 0;
  h.reuse = it->rz != nullptr ? 1 : 0, h.g_kind = it->o.g_kind;
  h.n = it->n, h.m = it->f->A->m, h.ld = it->f->A->ld, h.bytes = state_blob_bytes(it);
  h.gamma = it->gamma, h.f_x = it->f_x, h.g_z = it->g_z, h.res_inf = it->res_inf, h.dot_gr = it->dot_gr, h.res_sq = it->res_sq;
  h.beta = it->beta, h.f_z = it->f_z, h.f_z_upp = it->f_z_upp;
  h.seq_stepsize = it->seq_stepsize, h.seq_theta = it->seq_theta, h.seq_t = it->seq_t, h.seq_k = it->seq_k;
  h.sp_f = it->sp_f, h.sp_beta = it->sp_beta, h.spec_stepsize = it->spec_stepsize, h.spec_theta = it->spec_theta;
  h.spec_t = it->spec_t, h.spec_k = it->spec_k, h.a_passes = it->f->a_passes - it->passes0;
  h.flags = it->flags, h.n_backtracks = it->n_backtracks, h.rz_valid = it->rz_valid ? 1 : 0, h.sp_ready = it->sp_ready ? 1 : 0;
  h.sp_slot = it->sp_slot, h.r_current = it->sp_gen == it->f->r_gen ? 1 : 0;
  memcpy(host_blob, &h, sizeof(h));
  char* dst = (char*)host_blob + sizeof(h);
  void* ptrs[12];
  size_t nb[12];
  const int k = state_vectors(it, ptrs, nb);
  for (int i = 0; i < k; ++i) {
    if (nb[i]) PG_HIP(hipMemcpyAsync(dst, ptrs[i], nb[i], hipMemcpyDeviceToHost, c->stream));
    dst += nb[i];
  }
  PG_HIP(hipStreamSynchronize(c->stream));
  return PG_OK;
}

pg_status pg_iter_state_upload(pg_iter* it, const void* host_blob, int64_t bytes, pg_iter_scalars* out) {
  PG_REQUIRE(it != nullptr && host_blob != nullptr, "null argument");
  PG_REQUIRE(bytes >= (int64_t)sizeof(StateBlobHeader), "the blob is shorter than its header");
  StateBlobHeader h;
  memcpy(&h, host_blob, sizeof(h));
  PG_REQUIRE(h.magic == STATE_MAGIC && h.version == 1, "not a state blob of this library version");
  PG_REQUIRE(h.dtype == it->dtype && h.n == it->n && h.m == it->f->A->m && h.ld == it->f->A->ld,
             "the blob was saved for another problem size or element type");
  PG_REQUIRE(h.fast == it->o.fast && h.adaptive == (it->adaptive ? 1 : 0) && h.g_kind == it->o.g_kind,
             "the blob was saved by another iteration type (fast / adaptive / g)");
  // A solve that LEFT the single-sweep mode at run time (three bounded waits in a row, a refused launch: redo_with_two_sweeps)
  // saves single_sweep = 0 from an iterator that was created with it -- and a fresh iterator with the same options has it on.
  // The blob's layout follows from x_next being allocated, not from the flag (h.bytes is checked below), so such a blob is
  // taken and this iterator leaves the mode as well; the other direction (a blob WITH a speculative half into an iterator
  // without the seventh vector) has no place for it.
  PG_REQUIRE((h.single_sweep == (it->single_sweep ? 1 : 0) || (h.single_sweep == 0 && it->x_next != nullptr)) &&
                 h.reuse == (it->rz != nullptr ? 1 : 0),
             "the blob was saved by an iterator with other sweep options (single_sweep / reuse_residual / sharding)");
  PG_REQUIRE(h.bytes == state_blob_bytes(it) && bytes >= h.bytes, "the blob is truncated");
  if (h.single_sweep == 0 && it->single_sweep) {
    it->single_sweep = false;
    h.sp_ready = 0;
  }
  pg_ctx* c = it->ctx;
  const char* src = (const char*)host_blob + sizeof(h);
  void* ptrs[12];
  size_t nb[12];
  const int k = state_vectors(it, ptrs, nb);
  for (int i = 0; i < k; ++i) {
    if (nb[i]) PG_HIP(hipMemcpyAsync(ptrs[i], src, nb[i], hipMemcpyHostToDevice, c->stream));
    src += nb[i];
  }
  PG_HIP(hipStreamSynchronize(c->stream));  // the caller's blob may go away after this call
  it->gamma = h.gamma, it->f_x = h.f_x, it->g_z = h.g_z, it->res_inf = h.res_inf, it->dot_gr = h.dot_gr, it->res_sq = h.res_sq;
  it->beta = h.beta, it->f_z = h.f_z, it->f_z_upp = h.f_z_upp;
  it->seq_stepsize = h.seq_stepsize, it->seq_theta = h.seq_theta, it->seq_t = h.seq_t, it->seq_k = h.seq_k;
  it->sp_f = h.sp_f, it->sp_beta = h.sp_beta, it->spec_stepsize = h.spec_stepsize, it->spec_theta = h.spec_theta;
  it->spec_t = h.spec_t, it->spec_k = h.spec_k;
  it->flags = h.flags, it->n_backtracks = h.n_backtracks, it->rz_valid = h.rz_valid != 0, it->sp_ready = h.sp_ready != 0;
  it->sp_slot = h.sp_slot;
  // f at the speculative point lives in the scalar block too (PG_S_FNEXT + slot): a batched run after the resume takes f(x) of
  // its first iteration from there with the batch's one read-back, and this context's block has never held it.  (The block is
  // mapped host memory: the store below is what a kernel would have left; the stream is idle after the synchronisation above.)
  if (it->sp_ready) c->hscal[PG_S_FNEXT + it->sp_slot] = it->sp_f;
  it->fx_src = -1;
  it->defer_sync = false;
  it->f->r_gen++;  // f->r was rewritten
  it->sp_gen = h.r_current ? it->f->r_gen : it->f->r_gen - 1;
  it->passes0 = it->f->a_passes - h.a_passes;
  it->initialized = true;
  fill_scalars(it, out);
  return PG_OK;
}

pg_status pg_iter_state_view(pg_iter* it, pg_iter_state* out) {
  PG_REQUIRE(it != nullptr && out != nullptr, "null argument");
  out->x = it->x;
  out->grad_f_x = it->grad_f_x;
  out->y = it->y;
  out->z = it->z;
  out->res = it->res;
  out->z_prev = it->z_prev;
  out->grad_f_z = it->grad_f_z;
  return PG_OK;
}

}  // extern "C"
