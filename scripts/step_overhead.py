#!/usr/bin/env python3
"""Where a FastForwardBackward step's time goes on short sweeps: the Python iterator (`next(it)`, the reference's `iterate`),
the in-library loop with the stop rule checked every iteration (pg_iter_run) and every 16 (pg_iter_run_batched), against the
sweep kernel's own duration.  Usage: python scripts/step_overhead.py [m n]..."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proximalalgorithms.jl_amd as pa  # noqa: E402


def main():
    rest = [int(v) for v in sys.argv[1:]]
    shapes = [(rest[i], rest[i + 1]) for i in range(0, len(rest) - 1, 2)] or [(512, 1 << 20), (2048, 1 << 20), (16384, 131072), (16384, 1 << 20)]
    ctx = pa.get_context()
    for (m, n) in shapes:
        A = pa.HIPMatrix.synthetic(m, n, np.float32, seed=0)
        b = pa.HIPVector.from_numpy(np.random.default_rng(1).standard_normal(m).astype(np.float32))
        f = pa.LeastSquares(A, b)
        lam, Lf = np.float32(0.3), np.float32(4.0 * n / m)
        K = 200
        out = {}
        it = iter(pa.FastForwardBackwardIteration(f=f, g=pa.NormL1(lam), x0=np.zeros(n, np.float32), Lf=Lf))
        for _ in range(5):
            next(it)
        ctx.profile(True, kernels=("gemv_tn",))
        ctx.profile_reset()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(K):
            next(it)
        ctx.sync()
        out["python iterator"] = (time.perf_counter() - t0) / K
        prof = ctx.profile_read()
        ctx.profile(False)
        out["sweep kernel"] = prof["gemv_tn"][1] * 1e-3 / max(prof["gemv_tn"][0], 1)
        for name, ce in (("in-library loop, stop rule every iteration", 1), ("in-library loop, stop rule every 16", 16)):
            solver = pa.FastForwardBackward(maxit=K + 6, tol=0.0, device_loop=True, check_every=ce)
            solver(x0=np.zeros(n, np.float32), f=f, g=pa.NormL1(lam), Lf=Lf)  # warm
            ctx.sync()
            t0 = time.perf_counter()
            solver(x0=np.zeros(n, np.float32), f=f, g=pa.NormL1(lam), Lf=Lf)
            ctx.sync()
            out[name] = (time.perf_counter() - t0) / (K + 6)
        print(f"{m} x {n}: " + " ; ".join(f"{k} {v * 1e6:.1f} us" for k, v in out.items()), flush=True)
        del f, A


if __name__ == "__main__":
    main()
