#!/usr/bin/env python3
"""BASELINE config 4: PANOC (L-BFGS memory 5, adaptive step) on logistic loss + L1, m = 16384, n = 10^6, Float32.
Every iteration is 2 passes over A (A d and A' grad) plus 2 more in adaptive mode (A z and the kept A' grad at z is
skipped unless the quadratic branch needs it) plus one pair per rejected line-search trial.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=16384)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--loss", choices=["logistic", "sqdist"], default="logistic")
    ap.add_argument("--images", type=int, default=1, help="0: the explicit product A d (panoc.jl:180) instead of the L-BFGS image slab")
    ap.add_argument("--speculate", type=int, default=1, help="PANOCplus: 0 = two reads of A per iteration (round 4)")
    ap.add_argument("--pair-trials", type=int, default=1, help="ZeroFPR: 0 = one trial point of the line search per sweep (round 4)")
    ap.add_argument("--trio-trials", type=int, default=1, help="ZeroFPR: 0 = at most two trial points per sweep")
    ap.add_argument("--gamma-candidates", type=int, default=3, help="1 = one product A z per candidate of the step-size search (round 5)")
    ap.add_argument("--algo", choices=["panoc", "zerofpr", "panocplus", "ffb", "ffb-generic"], default="panoc",
                    help="ffb: FastForwardBackward (adaptive) on Composed(loss, A), engine 'composed' (one read of A per "
                         "iteration); ffb-generic: the same with separate GEMV passes")
    args = ap.parse_args()
    import proximalalgorithms.jl_amd as pa

    m, n, dtype = args.m, args.n, np.float32
    ctx = pa.get_context()
    A = pa.HIPMatrix.synthetic(m, n, dtype, seed=0)
    rng = np.random.default_rng(12345)
    k = max(1, n // 1000)
    x_true = np.zeros(n, dtype)
    x_true[rng.choice(n, size=k, replace=False)] = rng.standard_normal(k).astype(dtype)
    b = A.mul(pa.HIPVector.from_numpy(x_true))
    b.axpby_(1.0, b, 0.01, pa.HIPVector.from_numpy(rng.standard_normal(m).astype(dtype)))
    f = pa.LogisticLoss(b) if args.loss == "logistic" else pa.SquaredDistance(b)
    _, g0 = f.value_and_gradient(pa.HIPVector.zeros(m, dtype))
    lam = dtype(0.1) * A.mul_adjoint(g0).norm_inf()
    newton = {"panoc": "PANOCIteration", "zerofpr": "ZeroFPRIteration", "panocplus": "PANOCplusIteration"}
    if args.algo in newton:
        iteration = getattr(pa, newton[args.algo])(f=f, A=A, g=pa.NormL1(lam), x0=np.zeros(n, dtype), images=bool(args.images),
                                                   pair_trials=bool(args.pair_trials),
                                                   trio_trials=bool(args.trio_trials), speculate=bool(args.speculate),
                                                   gamma_candidates=args.gamma_candidates)
    else:
        iteration = pa.FastForwardBackwardIteration(f=pa.Composed(f, A), g=pa.NormL1(lam), x0=np.zeros(n, dtype),
                                                    engine="composed" if args.algo == "ffb" else "generic")
    it = iter(iteration)
    s = next(it)
    for _ in range(args.warmup):
        s = next(it)
    key = "A_passes" if args.algo in newton else "a_passes"
    p0 = iteration.counters.get(key, 0)
    ctx.profile(True)
    ctx.profile_reset()
    ctx.sync()
    t0 = time.perf_counter()
    taus = {}
    for _ in range(args.steps):
        s = next(it)
        float(s.res_inf if getattr(s, "res_inf", None) is not None else s.res.norm_inf()) / float(s.gamma) <= 1e-8
        if hasattr(s, "tau"):
            taus[str(float(s.tau))] = taus.get(str(float(s.tau)), 0) + 1
    ctx.sync()
    dt = time.perf_counter() - t0
    prof = ctx.profile_read()
    passes = iteration.counters.get(key, 0) - p0
    if args.algo == "ffb-generic":  # the generic engine does not count: 2 passes per gradient + 1 per line-search f(z)
        passes = 3 * args.steps
    gemv_ms = prof["gemv_n_partial"][1] + prof["gemv_t"][1] + prof["gemv_tn"][1]
    name = (args.algo.upper() if args.algo == "panoc" else {"zerofpr": "ZeroFPR", "panocplus": "PANOCplus"}.get(args.algo, "")) + \
        " iters/sec, %s + L1, m=%d n=%d f32, LBFGS(5), adaptive" + ("" if args.images else ", explicit A d") if args.algo in newton else \
        "FastForwardBackward (" + args.algo + ") iters/sec, %s + L1, m=%d n=%d f32, adaptive"
    out = {"metric": name % (args.loss, m, n),
           "value": args.steps / dt, "unit": "it/s", "n_gpus": 1, "steps": args.steps, "ms_per_step": 1e3 * dt / args.steps,
           "dtype": "f32", "data": "synthetic", "A_passes_per_step": passes / args.steps,
           "A_passes_with_first_iteration": iteration.counters.get(key, 0), "gamma_candidates_ahead": iteration.counters.get("gamma_candidates_ahead", 0),
           "roofline": {"bound": "hbm", "kernels": "gemv_n_partial + gemv_t + gemv_tn",
                        "achieved": passes * m * n * 4 / (gemv_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": passes * m * n * 4 / (gemv_ms * 1e-3) / 1e9 / 8000.0,
                        "gemv_time_fraction_of_step": gemv_ms * 1e-3 / dt},
           "whole_iteration_GBps": passes * m * n * 4 / dt / 1e9, "accepted_tau_histogram": taus,
           "pair_sweeps": int(getattr(s, "pair_sweeps", 0)), "trio_sweeps": int(getattr(s, "trio_sweeps", 0)),
           "final": {"gamma": float(s.gamma), "tau": float(getattr(s, "tau", 0.0)), "res_inf_over_gamma": float(s.res.norm_inf()) / float(s.gamma)}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
