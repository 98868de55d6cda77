#!/usr/bin/env python3
"""PANOC / ZeroFPR / PANOCplus run TO THE STOPPING RULE on the device and in the oracle at BASELINE config 4's column length
(logistic + L1, 16384 x n, Float32, L-BFGS(5), adaptive step, image slab on): iterations taken and the final objectives
(north_star's "final objective within 1e-6 rel", asserted unconditionally by
tests/test_gpu_parity.py::test_newton_family_final_objective_at_config4_column_length).  One JSON line per (algorithm, tol).
    python tests/tools/newton_stop_rule.py --n 32768 --tols 1e-3,3e-4,1e-4 [--algs PANOC,ZeroFPR,PANOCplus] [--no-oracle]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def problem(pa, m, n, seed=5):
    dtype = np.float32
    ctx = pa.get_context()
    A_d = pa.HIPMatrix.synthetic(m, n, dtype, seed=seed, ctx=ctx)
    rng = np.random.default_rng(12345)
    x_true = np.zeros(n, dtype)
    x_true[rng.choice(n, size=n // 1000, replace=False)] = rng.standard_normal(n // 1000).astype(dtype)
    b_d = A_d.mul(pa.HIPVector.from_numpy(x_true, ctx))
    b_d.axpby_(1.0, b_d, 0.01, pa.HIPVector.from_numpy(rng.standard_normal(m).astype(dtype), ctx))
    _, g0 = pa.LogisticLoss(b_d).value_and_gradient(pa.HIPVector.zeros(m, dtype, ctx))
    lam = dtype(0.1) * A_d.mul_adjoint(g0).norm_inf()
    return A_d, b_d, lam


def objective(A, b64, lam):
    def obj(z):
        nz = np.flatnonzero(z)
        t = A[:, nz].astype(np.float64) @ z[nz].astype(np.float64) - b64
        return float(np.sum(np.log1p(np.exp(-t)))) + float(lam) * float(np.sum(np.abs(z.astype(np.float64))))

    return obj


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=16384)
    ap.add_argument("--n", type=int, default=32768)
    ap.add_argument("--tols", default="1e-3,3e-4,1e-4")
    ap.add_argument("--algs", default="PANOC,ZeroFPR,PANOCplus")
    ap.add_argument("--maxit", type=int, default=400)
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--refresh-every", type=int, default=0)
    args = ap.parse_args()
    import proximalalgorithms.jl_amd as pa
    from oracle import proxgrad_oracle as o

    A_d, b_d, lam = problem(pa, args.m, args.n)
    A, b = A_d.numpy(), b_d.numpy()
    obj = objective(A, b.astype(np.float64), lam)
    x0 = np.zeros(args.n, np.float32)
    oname = {"PANOC": "panoc", "ZeroFPR": "zerofpr", "PANOCplus": "panocplus"}
    for alg in args.algs.split(","):
        for tol in (float(t) for t in args.tols.split(",")):
            kw = {} if alg == "ZeroFPR" or not args.refresh_every else {"refresh_every": args.refresh_every}
            t0 = time.perf_counter()
            zg, kg = getattr(pa, alg)(tol=tol, maxit=args.maxit, **kw)(x0=x0, f=pa.LogisticLoss(b_d), A=A_d, g=pa.NormL1(lam))
            tg = time.perf_counter() - t0
            zg = zg.numpy() if hasattr(zg, "numpy") else np.asarray(zg)
            Fg = obj(zg)
            line = {"alg": alg, "tol": tol, "k_gpu": int(kg), "F_gpu": Fg, "gpu_s": round(tg, 2)}
            if not args.no_oracle:
                t0 = time.perf_counter()
                zo, ko = getattr(o, oname[alg])(tol=tol, maxit=args.maxit, x0=x0, f=o.LogisticLoss(b), A=A, g=o.NormL1(lam))
                line.update(k_cpu=int(ko), F_cpu=obj(zo), cpu_s=round(time.perf_counter() - t0, 2))
                line["rel_diff"] = abs(Fg - line["F_cpu"]) / abs(line["F_cpu"])
            print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
