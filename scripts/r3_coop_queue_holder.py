#!/usr/bin/env python3
"""A process that runs one small solve -- with a cooperative launch (`coop`: the one-launch cooperative solver) or with plain
launches only (`plain`) -- and then sits idle for N seconds, still holding its HIP queues.  Used to show that an IDLE process
which has used a cooperative launch slows the cooperative team sweep of ANOTHER process on the same device to 0.45 of its rate
(profiles/r3_team_coop_vs_plain.md):
    python scripts/r3_coop_queue_holder.py coop 40 &  sleep 12;  python bench.py --m 131072 --n 131072 --no-also --no-cpu-baseline"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import proximalalgorithms.jl_amd as pa
from oracle import proxgrad_oracle as o
mode = sys.argv[1]
ctx = pa.get_context()
if mode == "coop":
    As, bs, _ = o.synthetic_lasso(256, 400, seed=2, dtype=np.float64)
    pa.FastForwardBackward(tol=1e-6, maxit=50, device_loop=True)(x0=np.zeros(400), f=pa.LeastSquares(As, bs), g=pa.NormL1(0.01), Lf=4.0)
elif mode == "plain":
    As, bs, _ = o.synthetic_lasso(2048, 4000, seed=2, dtype=np.float32)
    pa.FastForwardBackward(tol=1e-6, maxit=50)(x0=np.zeros(4000, np.float32), f=pa.LeastSquares(As, bs), g=pa.NormL1(0.01), Lf=4.0)
ctx.sync()
print("holder", mode, "idle now", flush=True)
time.sleep(float(sys.argv[2]))
