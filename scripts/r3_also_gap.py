#!/usr/bin/env python3
"""Why does config 2 run ~4 % slower inside the default line's also[] (after the 64 GiB headline matrix was freed) than in
a process of its own?  One process: config 2 first, then the headline, then config 2 again (re-allocated each time, more
warm-up, more steps), printing it/s and the address of A each time.
    python scripts/r3_also_gap.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
import proximalalgorithms.jl_amd as pa

args = bench.parse_args(["--no-cpu-baseline", "--no-also"])
ctx = pa.get_context(0)
D = bench.Dist(1, 0, 0, "nccl", "torch", False, False)
D.beat = lambda: None


def rec(m, n, steps, warm, tag):
    P = bench.setup_lasso(pa, ctx, D, m, n, np.float32, 0, "none", "fixed")
    r = bench.run_ffb(pa, ctx, D, P, "fixed", "one", steps, warm, "main")
    print("%-34s %8.1f it/s  frac %.4f  A at 0x%x" % (tag, r["value"], r["roofline"]["frac"], P["A"].info()["ptr"]), flush=True)
    return P


m2, n2 = bench.WORKLOADS["config2"]
for i in range(2):
    rec(m2, n2, 50, 5, "config2 fresh process #%d" % i)
rec(m2, n2, 300, 50, "config2, 50 warm-up 300 steps")
P = rec(16384, 1 << 20, 20, 5, "headline")
del P
import gc

gc.collect()
for i in range(3):
    rec(m2, n2, 50, 5, "config2 after the headline #%d" % i)
rec(m2, n2, 300, 50, "config2 after, 50 warm-up 300")

# one iterator, one set of addresses: 12 consecutive windows of 50 steps -- is the spread temporal?
import time

P = bench.setup_lasso(pa, ctx, D, m2, n2, np.float32, 0, "none", "fixed")
iteration = pa.FastForwardBackwardIteration(f=P["f"], g=pa.NormL1(P["lam"]), x0=P["zero_n"], Lf=P["Lf"])
it = iter(iteration)
for _ in range(10):
    next(it)
rates = []
for w in range(12):
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(50):
        next(it)
    ctx.sync()
    rates.append(50 / (time.perf_counter() - t0))
    if w == 5:
        time.sleep(2.0)  # an idle gap: does the rate after it differ?
print("same iterator, 12 windows of 50 steps (2 s idle after the 6th): " + " ".join("%.0f" % r for r in rates), flush=True)
