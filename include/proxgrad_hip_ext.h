/*
 * proxgrad_hip_ext.h -- entry points of libproxgrad_hip.so OUTSIDE the hot-path scope of SURVEY.md section 8.
 *
 * proxgrad_hip.h is the drop-in boundary for the ForwardBackward / FastForwardBackward path and its "next" rows
 * (DouglasRachford, PANOC / ZeroFPR / PANOCplus with L-BFGS, the device-resident loops).  What is declared here serves
 * other algorithms of the reference on the same kernels -- Davis-Yin and AFBA bodies replayed as graphs, the Broyden
 * operator, the fault-injection hook of the tests -- and is kept apart so that the boundary header matches section 8(b)'s table.
 * Same conventions as proxgrad_hip.h.
 */
#ifndef PROXGRAD_HIP_EXT_H
#define PROXGRAD_HIP_EXT_H

#include "proxgrad_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ stream capture ---- */
/* Stream capture for launch-bound iteration bodies ("capture launch-bound inner loops in hipGraphs"): between _begin and
 * _end every entry point of this library called on the context RECORDS its kernels into a graph instead of running
 * them; scalar outputs (double* ..._out) are not meaningful for calls made during the capture.  The recorded body --
 * e.g. one Base.iterate of AFBA / DavisYin / DouglasRachford (primal_dual.jl:176-209, davis_yin.jl:73-83,
 * douglas_rachford.jl:57-63), whose step sizes are constants -- is then replayed with ONE launch per iteration.
 * Requirements: the context owns a non-default stream, no collective attached, every workspace the body needs was
 * allocated by a previous (uncaptured) run of the same body.  _end with out == NULL aborts and discards the capture. */
typedef struct pg_graph pg_graph;
pg_status pg_ctx_capture_begin(pg_ctx* ctx);
pg_status pg_ctx_capture_end(pg_ctx* ctx, pg_graph** graph_out);
pg_status pg_graph_launch(pg_graph* graph);
pg_status pg_graph_destroy(pg_graph* graph);

/* ------------------------------------------------------------------ Broyden rank-one update ---- */
/* A += alpha * u * w'  (u: m-vector, w: n-vector, device): the `L.H .+= (s - Hy) / dot(...) * sH` rank-one update of
 * the Broyden operator, src/accel/broyden.jl:18-28 */
pg_status pg_mat_rank1_update(pg_mat* A, double alpha, const void* u, const void* w);

/* ------------------------------------------------------------------ Davis-Yin single sweep ---- */
/* One Davis-Yin iteration (davis_yin.jl:73-83: prox!(xg, g, z); grad f(xg); z_half = 2 xg - z - gamma grad; prox!(xh, h,
 * z_half); res = xh - xg; z += lambda res) for f = loss o A in ONE read of A.  Input: r = grad loss(A xg) (m-vector), xg, z.
 * Output per column: grad = A' r, z_half, xh = prox_{gamma h}(z_half), res, z_next = z + relax * res, and already the NEXT
 * iteration's xg_next = prox_{gamma g}(z_next) with its image A_xg_next = A xg_next.  g_kind / h_kind in {PG_G_ZERO,
 * PG_G_NORML1 (p0 = lam), PG_G_INDBOX (p0 = lo, p1 = hi), PG_G_SQRNORML2 (p0 = lam)}.  scalars_out (host, may be NULL) =
 * { 0, norm(res, Inf), dot(grad, res), norm(res)^2 }.  Same shape limits as pg_mat_fused_tn. */
pg_status pg_mat_fused_dys(pg_mat* A, const void* r, const void* xg, const void* z, double gamma, double relax,
                           int32_t g_kind, double g_p0, double g_p1, int32_t h_kind, double h_p0, double h_p1, void* grad,
                           void* z_half, void* xh, void* res, void* z_next, void* xg_next, void* A_xg_next,
                           double* scalars_out);

/* ------------------------------------------------------------------ fault injection (tests) ---- */
/* The kth_launch-th long-column (team) sweep launched on this context from now on (1-based; 0 switches the hook off) fails:
 * kind 0 -- one workgroup of one team is never started, its team-mates run into their bounded wait and the step is redone with
 * two sweeps (PG_FLAG_SWEEP_FALLBACK); kind 1 -- the launch is refused (PG_ERR_UNSUPPORTED), as a cooperative launch that
 * does not fit next to other work would be (a row-team sweep, pg_ctx_set_row_team, is a plain launch that is never refused:
 * there kind 1 acts like kind 0).  Nothing in the library reads the environment for this.
 * kind 4 (profiling): kth_launch != 0 makes a "team" of ONE device a row team from the next pg_ctx_set_row_team(ctx, 1, 0, {own
 * inbox}, ...) on: the sweep posts to and polls its own inbox, so the whole instruction stream of the exchange runs with the
 * kernel ALONE on the device -- what rocprofv3 --pmc needs, which serialises kernels (two ranks' sweeps would wait for each other
 * until their bounded wait expires). */
pg_status pg_ctx_test_team_fault(pg_ctx* ctx, int32_t kth_launch, int32_t kind);
/* kind 2 -- LATENCY INJECTOR of the row-team sweep (the hand-off of benchmark/benchmarks.jl:16's partial sums between
 * devices): from now on the sweep is launched in its DELAY form and a step's granules are accepted by their consumers only
 * kth_launch NANOSECONDS after they were stored (a stamp of the device's constant 100 MHz clock travels with them), i.e. the
 * hand-off takes max(what it takes on this device, that long).  All members of the team must share one device (one clock):
 * the one-GPU test set-up.  0 ns: the injector's own cost.  kind 3 switches it off again.  Only the geometries the latency
 * sweep uses are instantiated (PG_ERR_UNSUPPORTED otherwise). */
/* Slack read-out of the injector: clock ticks (10 ns) between the stamp of a step's granules (member 0's) and their use, summed over
 * the wave-steps counted, since pg_ctx_set_row_team -- how much later the granules could have come without a wave waiting. */
pg_status pg_ctx_test_team_slack(pg_ctx* ctx, int64_t* ticks, int64_t* wave_steps);

#ifdef __cplusplus
}
#endif
#endif /* PROXGRAD_HIP_EXT_H */
