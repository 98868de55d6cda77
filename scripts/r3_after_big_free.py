#!/usr/bin/env python3
"""After the 64 GiB headline matrix is freed: how long until config 2 runs at its own rate?  (The default bench line
measures config 2 right after that free.)  Windows of 20 steps, printed with the time since the free.
    python scripts/r3_after_big_free.py"""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
import proximalalgorithms.jl_amd as pa

ctx = pa.get_context(0)
D = bench.Dist(1, 0, 0, "nccl", "torch", False, False)
D.beat = lambda: None
m2, n2 = bench.WORKLOADS["config2"]
for trial in range(2):
    big = pa.HIPMatrix.synthetic(16384, 1 << 20, np.float32, seed=0, ctx=ctx)
    ctx.sync()
    del big
    gc.collect()
    t_free = time.perf_counter()
    P = bench.setup_lasso(pa, ctx, D, m2, n2, np.float32, 0, "none", "fixed")
    iteration = pa.FastForwardBackwardIteration(f=P["f"], g=pa.NormL1(P["lam"]), x0=P["zero_n"], Lf=P["Lf"])
    it = iter(iteration)
    out = []
    while time.perf_counter() - t_free < 4.0:
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(20):
            next(it)
        ctx.sync()
        t1 = time.perf_counter()
        out.append((t1 - t_free, 20 / (t1 - t0)))
    print("trial %d: " % trial + " ".join("%.2fs:%.0f" % o for o in out[::4]), flush=True)
    del it, iteration, P
    gc.collect()
