#!/usr/bin/env python3
"""Davis-Yin on an elastic-net LASSO (f = ||A x - b||^2 / 2, g = lam1 ||.||_1, h = lam2/2 ||.||^2) at the headline size:
iterations/sec with the single sweep (pg_mat_fused_dys: one read of A per iteration) and with the reference's statement
order (A x and A' r as separate products).  Prints one JSON line per mode."""
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
    for _ in range(30):
        A.mul(v, u); A.mul_adjoint(u, v); nrm = float(v.norm()); v.axpby_(1.0 / nrm, v)
    Lf = 1.1 * nrm
    for single in (True, False):
        it = pa.DavisYinIteration(x0=np.zeros(n, dtype), f=pa.LeastSquares(A, b), g=pa.NormL1(lam), h=pa.SqrNormL2(dtype(0.1)),
                                  Lf=Lf, single_sweep=single)
        gen = iter(it)
        for _ in range(3): s = next(gen)
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(args.steps):
            s = next(gen)
            float(s.res_inf if getattr(s, "res_inf", None) is not None else s.res.norm_inf()) <= 1e-8
        ctx.sync(); dt = (time.perf_counter() - t0) / args.steps
        reads = 1 if single else 2
        print(json.dumps({"metric": "DavisYin iters/sec, elastic-net LASSO m=%d n=%d f32" % (m, n), "single_sweep": single,
                          "value": 1 / dt, "unit": "it/s", "ms_per_step": 1e3 * dt, "reads_of_A_per_step": reads,
                          "frac_of_8TBps": reads * m * n * 4 / dt / 8e12}), flush=True)

if __name__ == "__main__":
    main()
