#!/usr/bin/env python3
"""The first ~100 iterations on a matrix set up after another one was freed run 1-4 % slower (scripts/r3_alloc_spread.py).
Time- or step-bound?  Per problem: optional idle time after the set-up, then windows of 10 steps.
    python scripts/r3_first_window.py [config2|headline]"""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
import proximalalgorithms.jl_amd as pa

which = sys.argv[1] if len(sys.argv) > 1 else "config2"
m, n = bench.WORKLOADS[which]
ctx = pa.get_context(0)
D = bench.Dist(1, 0, 0, "nccl", "torch", False, False)
D.beat = lambda: None
for trial, idle in enumerate((0.0, 0.0, 1.0, 0.0, 3.0, 0.0)):
    t_s = time.perf_counter()
    P = bench.setup_lasso(pa, ctx, D, m, n, np.float32, 0, "none", "fixed")
    t_setup = time.perf_counter() - t_s
    time.sleep(idle)
    iteration = pa.FastForwardBackwardIteration(f=P["f"], g=pa.NormL1(P["lam"]), x0=P["zero_n"], Lf=P["Lf"])
    it = iter(iteration)
    out = []
    for w in range(16):
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(10):
            next(it)
        ctx.sync()
        out.append(10 / (time.perf_counter() - t0))
    print("problem %d (set-up %.2f s, idle %.0f s): " % (trial, t_setup, idle) + " ".join("%.0f" % r for r in out), flush=True)
    del it, iteration, P
    gc.collect()
