#!/usr/bin/env python3
"""Where each form of the FFB driver loop wins (it/s over 400 iterations, fixed and adaptive step): the single-workgroup
kernel (pg_iter_run_small), the cooperative multi-workgroup kernel (pg_iter_run_coop) and the host-driven streaming kernels
(pg_iter_run, batched by 16 when the step is fixed).  Feeds the dispatch thresholds of algorithm.py."""
import os, sys, time, gc, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import proximalalgorithms.jl_amd as pa
from oracle import proxgrad_oracle as o
for (m, n, dt) in ((100, 200, np.float64), (130, 300, np.float64), (300, 700, np.float32), (500, 1000, np.float32), (700, 1500, np.float64), (1000, 2000, np.float32), (1000, 2000, np.float64), (1500, 3000, np.float32), (2000, 4000, np.float32), (3000, 6000, np.float32), (4000, 8000, np.float32)):
    A, b, _ = o.synthetic_lasso(m, n, seed=0, dtype=dt)
    lam = dt(0.1 * np.max(np.abs(A.T @ b))); Lf = dt(1.05 * np.linalg.norm(A.astype(np.float64), 2) ** 2)
    f, g = pa.LeastSquares(A, b), pa.NormL1(lam)
    row = []
    for kw in (dict(Lf=Lf), {}):
        res = {}
        for name in ("coop", "coop", "host", "host1", "small"):
            if name == "small" and m * n > 2**20: continue
            it = pa.FastForwardBackwardIteration(f=f, g=g, x0=np.zeros(n, dt), **kw); next(iter(it)); gc.collect()
            t0 = time.perf_counter()
            if name == "coop": k, _ = it._fused.run_coop(1, 401, 0.0, 0)
            elif name == "small": k, _ = it._fused.run_small(1, 401, 0.0)
            elif name == "host1": k, _ = it._fused.run(1, 401, 0.0)  # single-sweep steps, one sync per iteration
            else: k, _ = it._fused.run(1, 401, 0.0, check_every=(16 if kw else 1))
            res[name] = (k - 1) / (time.perf_counter() - t0)
        row.append({k_: round(v) for k_, v in res.items()})
    print(m, n, np.dtype(dt).name, "fixed", row[0], "adaptive", row[1], flush=True)
