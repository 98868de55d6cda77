#!/usr/bin/env python3
"""Chambolle-Pock (AFBA with theta = 2) on LASSO  min lam ||x||_1 + ||A x - b||^2 / 2  at the headline size (L = A, 64 GiB
Float32): iterations/sec with the single sweep (one read of A per iteration) and with the reference's statement order
(L'y and L(2 xbar - x) as separate products).  Prints one JSON line per mode."""
import argparse, json, math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proximalalgorithms.jl_amd as pa

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=16384)
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    m, n, dtype = args.m, args.n, np.float32
    ctx = pa.get_context()
    A = pa.HIPMatrix.synthetic(m, n, dtype, seed=0)
    rng = np.random.default_rng(12345)
    k = max(1, n // 1000)
    x_true = np.zeros(n, dtype)
    x_true[rng.choice(n, size=k, replace=False)] = rng.standard_normal(k).astype(dtype)
    b = A.mul(pa.HIPVector.from_numpy(x_true))
    lam = dtype(0.1) * A.mul_adjoint(b).norm_inf()
    v, u = pa.HIPVector.zeros(n, dtype).fill_(1.0 / math.sqrt(n)), pa.HIPVector.empty(m, dtype)
    nrm = 1.0
    for _ in range(30):  # ||A||^2 by power iteration (setup, untimed)
        A.mul(v, u); A.mul_adjoint(u, v); nrm = float(v.norm()); v.axpby_(1.0 / nrm, v)
    opn = 1.05 * math.sqrt(nrm)
    for single in (True, False):
        it = pa.ChambollePockIteration(x0=np.zeros(n, dtype), y0=np.zeros(m, dtype), g=pa.NormL1(lam), h=pa.SquaredDistance(b),
                                       L=A, opnorm_L=opn, single_sweep=single)
        gen = iter(it)
        for _ in range(3): s = next(gen)
        p0 = it.counters["L_passes"]
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(args.steps):
            s = next(gen)
            float(s.FPR_x.norm_inf()) + float(s.FPR_y.norm_inf()) <= 1e-8   # the stop rule, evaluated
        ctx.sync(); dt = time.perf_counter() - t0
        passes = (it.counters["L_passes"] - p0) / args.steps
        print(json.dumps({"metric": "ChambollePock iters/sec, LASSO m=%d n=%d f32, L = A" % (m, n), "single_sweep": single,
                          "value": args.steps / dt, "unit": "it/s", "ms_per_step": 1e3 * dt / args.steps, "reads_of_A_per_step": passes,
                          "whole_iteration_GBps": passes * m * n * 4 / (dt / args.steps) / 1e9,
                          "frac_of_8TBps": passes * m * n * 4 / (dt / args.steps) / 8e12}), flush=True)

if __name__ == "__main__":
    main()
