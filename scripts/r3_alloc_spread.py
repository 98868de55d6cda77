#!/usr/bin/env python3
"""Where does the run-to-run spread of one shape come from?  Six problems set up one after the other in one process (the
matrix freed and re-allocated each time), two iterators on each, two windows of steps per iterator.
    python scripts/r3_alloc_spread.py [config2|headline] [keep]      keep: the previous matrix stays allocated (new addresses)"""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
import proximalalgorithms.jl_amd as pa

which = sys.argv[1] if len(sys.argv) > 1 else "config2"
keep_prev = len(sys.argv) > 2 and sys.argv[2] == "keep"
m, n = bench.WORKLOADS[which]
steps = 100 if which == "config2" else 15
ctx = pa.get_context(0)
D = bench.Dist(1, 0, 0, "nccl", "torch", False, False)
D.beat = lambda: None
prev = None
for trial in range(6):
    P = bench.setup_lasso(pa, ctx, D, m, n, np.float32, 0, "none", "fixed")
    if not keep_prev:
        prev = None
    out = []
    for k in range(2):
        iteration = pa.FastForwardBackwardIteration(f=P["f"], g=pa.NormL1(P["lam"]), x0=P["zero_n"], Lf=P["Lf"])
        it = iter(iteration)
        for _ in range(8):
            next(it)
        for w in range(2):
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                next(it)
            ctx.sync()
            out.append(steps / (time.perf_counter() - t0))
        del it, iteration
    print("problem %d: A at 0x%x  iterator 1: %.1f %.1f  iterator 2: %.1f %.1f" % (trial, P["A"].info()["ptr"], *out), flush=True)
    prev = P if keep_prev else None
    del P
    gc.collect()
