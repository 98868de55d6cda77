#!/usr/bin/env python3
"""Does the placement of the iteration's state vectors move the sweep's rate?  (scripts/r3_also_gap.py: the same matrix at
the same address runs 777..820 it/s from one iterator to the next, +-0.2 % within one.)  One matrix; iterators created with
the state vectors staggered (PG_ITER_VEC_SKEW) or shifted (PG_ITER_BASE_SKEW), three interleaved rounds.
    python scripts/r3_slab_layout.py [config2|headline]"""
import os

os.environ.setdefault("PG_TUNE", "1")  # the library reads its tuning variables only when this is set

import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
import proximalalgorithms.jl_amd as pa

which = sys.argv[1] if len(sys.argv) > 1 else "config2"
m, n = bench.WORKLOADS[which]
steps = 100 if which == "config2" else 20
ctx = pa.get_context(0)
D = bench.Dist(1, 0, 0, "nccl", "torch", False, False)
D.beat = lambda: None
P = bench.setup_lasso(pa, ctx, D, m, n, np.float32, 0, "none", "fixed")
a_ptr = P["A"].info()["ptr"]
settings = [(0, 0), (256, 0), (512, 0), (1024, 0), (2048, 0), (4096, 0), (8192, 0), (16384, 0), (65536, 0), (4096 + 256, 0), (65536 + 4096 + 256, 0),
            (0, 256), (0, 4096), (0, 65536), (0, 1 << 20), (0, (1 << 20) + 4096 + 256)]
res = {s: [] for s in settings}
keep = []
for rnd in range(3):
    for s in settings:
        os.environ["PG_ITER_VEC_SKEW"], os.environ["PG_ITER_BASE_SKEW"] = str(s[0]), str(s[1])
        iteration = pa.FastForwardBackwardIteration(f=P["f"], g=pa.NormL1(P["lam"]), x0=P["zero_n"], Lf=P["Lf"])
        it = iter(iteration)
        for _ in range(8):
            next(it)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            st = next(it)
        ctx.sync()
        rate = steps / (time.perf_counter() - t0)
        xp = iteration._fused.view()["x"].ptr
        res[s].append((rate, xp))
        if rnd == 0:
            keep.append(iteration)  # keep the first round's slabs alive: later rounds land elsewhere
print("# %s %d x %d, A at 0x%x; rate it/s (x at, relative to A, mod 2 MiB) per round" % (which, m, n, a_ptr))
for s in settings:
    print("vec_skew %7d base_skew %8d : " % s + "  ".join("%6.1f (0x%x, %+d KiB mod 2Mi)" % (r, xp, ((xp - a_ptr) % (2 << 20)) // 1024) for r, xp in res[s]))
