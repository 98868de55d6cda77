#!/usr/bin/env python3
"""BASELINE config 3: DouglasRachford on a box-constrained QP with diagonal Hessian, n = 10^7, Float32.
One fused HBM sweep per iteration (pg_dr_step).  Prints one JSON line (not the driver's bench: see bench.py)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    import proximalalgorithms.jl_amd as pa
    from oracle import proxgrad_oracle as o

    n, dtype = args.n, np.float32
    rng = np.random.default_rng(0)
    d = (0.1 + rng.random(n)).astype(dtype)
    q = rng.standard_normal(n).astype(dtype)
    x0 = np.zeros(n, dtype)
    lo, hi, gamma = dtype(-0.5), dtype(0.25), dtype(1.0)
    ctx = pa.get_context()
    out = {"metric": "DouglasRachford iters/sec, box QP n=%d f32" % n, "unit": "it/s", "n_gpus": 1, "dtype": "f32",
           "data": "synthetic", "modes": {}}
    for name, mat, nstreams in (("x_y_only", False, 5), ("full_state", True, 8)):
        it = iter(pa.DouglasRachfordIteration(f=pa.SeparableQuadratic(d, q), g=pa.IndBox(lo, hi), x0=x0, gamma=gamma,
                                              materialize=mat))
        for _ in range(args.warmup):
            s = next(it)
        ctx.profile(True)
        ctx.profile_reset()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            s = next(it)
            float(s.res_inf) / float(gamma) <= 1e-8  # the stop rule, evaluated every iteration like the driver loop
        ctx.sync()
        dt = time.perf_counter() - t0
        cnt, ms = ctx.profile_read()["dr_step"]
        ctx.profile(False)
        bytes_iter = nstreams * n * 4
        out["modes"][name] = {"value": args.steps / dt, "ms_per_step": 1e3 * dt / args.steps,
                              "algorithmic_bytes_per_iter": bytes_iter,
                              "roofline": {"bound": "hbm", "kernel": "dr_step", "avg_launch_ms": ms / cnt,
                                           "achieved": bytes_iter / (ms / cnt * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                           "frac": bytes_iter / (ms / cnt * 1e-3) / 1e9 / 8000.0},
                              "whole_iteration_GBps": bytes_iter * args.steps / dt / 1e9,
                              "res_inf": float(s.res_inf)}
    # driver loop inside the library, K iterations per HBM sweep (pg_dr_run); tol = 0 so that exactly `steps` run
    for block in (8, 16, 32, 64):
        steps = max(args.steps, 10 * block) // block * block
        itn = pa.DouglasRachfordIteration(f=pa.SeparableQuadratic(d, q), g=pa.IndBox(lo, hi), x0=x0, gamma=gamma,
                                          materialize=False)
        itn.device_run(2 * block, 0.0, block)  # warm-up
        ctx.profile(True)
        ctx.profile_reset()
        ctx.sync()
        t0 = time.perf_counter()
        s, k = itn.device_run(steps, 0.0, block)
        ctx.sync()
        dt = time.perf_counter() - t0
        cnt, ms = ctx.profile_read()["dr_step"]
        ctx.profile(False)
        assert k == steps
        bytes_launch = 5 * n * 4  # x, d, q in; x, y out -- per K iterations
        out["modes"]["device_loop_block%d" % block] = {
            "value": steps / dt, "ms_per_step": 1e3 * dt / steps, "iterations": steps, "launches": cnt,
            "algorithmic_bytes_per_launch": bytes_launch,
            "roofline": {"bound": "hbm", "kernel": "dr_block<%d>" % block, "avg_launch_ms": ms / cnt,
                         "achieved": bytes_launch / (ms / cnt * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": bytes_launch / (ms / cnt * 1e-3) / 1e9 / 8000.0},
            "res_inf": float(s.res_inf)}
    if not args.no_cpu_baseline:
        ito = iter(o.DouglasRachfordIteration(f=o.SeparableQuadratic(d, q), g=o.IndBox(lo, hi), x0=x0, gamma=gamma))
        next(ito)
        t0 = time.perf_counter()
        k = 5
        for _ in range(k):
            so = next(ito)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": k / dt, "unit": "it/s", "cores": 1, "kind": "port",
                               "sample": f"oracle DR (numpy, elementwise ops are single-threaded like Julia broadcasts), n={n}, {k} it"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
