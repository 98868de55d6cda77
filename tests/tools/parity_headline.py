#!/usr/bin/env python3
"""Parity AT THE HEADLINE SIZE (m = 16384, n = 2^20, Float32, 64 GiB): the GPU engine against the CPU restatement of
the reference on identical inputs.  The device matrix is downloaded to host memory (the GPU box has the RAM for it),
the oracle runs the reference's op sequence with OpenBLAS on the host cores, and iterates / step sizes / objectives are
compared.  Objectives are evaluated in float64 from the non-zero columns of z (z is sparse after the soft threshold),
so the comparison is not limited by a float32 evaluation.  Prints one JSON document."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def objective64(A, b, lam, z):
    nz = np.flatnonzero(z)
    r = A[:, nz].astype(np.float64) @ z[nz].astype(np.float64) - b.astype(np.float64)
    return 0.5 * float(r @ r) + float(lam) * float(np.sum(np.abs(z.astype(np.float64)))), int(nz.size)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=16384)
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--fixed-steps", type=int, default=30)
    ap.add_argument("--adaptive-steps", type=int, default=16)
    args = ap.parse_args()
    import proximalalgorithms.jl_amd as pa
    from oracle import proxgrad_oracle as o

    m, n, dtype = args.m, args.n, np.float32
    ctx = pa.get_context()
    t0 = time.perf_counter()
    A_d = pa.HIPMatrix.synthetic(m, n, dtype, seed=0, ctx=ctx)
    rng = np.random.default_rng(12345)
    k = max(1, n // 1000)
    x_true = np.zeros(n, dtype)
    x_true[rng.choice(n, size=k, replace=False)] = rng.standard_normal(k).astype(dtype)
    noise = np.random.default_rng(54321).standard_normal(m).astype(dtype)
    b_d = A_d.mul(pa.HIPVector.from_numpy(x_true, ctx))
    b_d.axpby_(1.0, b_d, 0.01, pa.HIPVector.from_numpy(noise, ctx))
    f_d = pa.LeastSquares(A_d, b_d)
    zero = pa.HIPVector.zeros(n, dtype, ctx)
    _, g0 = f_d.value_and_gradient(zero)
    lam = dtype(0.1) * g0.norm_inf()
    v = pa.HIPVector.zeros(n, dtype, ctx).fill_(1.0 / np.sqrt(n))
    f0 = pa.LeastSquares(A_d, pa.HIPVector.zeros(m, dtype, ctx))
    w = v.similar()
    nrm = dtype(1)
    for _ in range(30):
        f0.value_and_gradient(v, out=w)
        nrm = w.norm()
        v.axpby_(1.0 / float(nrm), w)
    Lf = dtype(1.1) * nrm
    del f0
    t_setup = time.perf_counter() - t0
    t0 = time.perf_counter()
    A = A_d.numpy()  # 64 GiB, column-major like the Julia Matrix
    b = b_d.numpy()
    t_dl = time.perf_counter() - t0
    out = {"m": m, "n": n, "dtype": "f32", "lambda": float(lam), "Lf": float(Lf), "setup_s": round(t_setup, 1),
           "download_s": round(t_dl, 1), "host_matrix_GiB": round(A.nbytes / 2**30, 1)}
    x0 = np.zeros(n, dtype)

    # ---------------- fixed step: iterate sequence (SURVEY 8(c) parity definition (i)) ----------------
    checkpoints = sorted({1, 2, 5, 10, 20, args.fixed_steps} & set(range(1, args.fixed_steps + 1)) | {args.fixed_steps})
    it_g = iter(pa.FastForwardBackwardIteration(f=f_d, g=pa.NormL1(lam), x0=x0, Lf=Lf))
    it_c = iter(o.FastForwardBackwardIteration(f=o.LeastSquares(A, b), g=o.NormL1(lam), x0=x0, Lf=Lf))
    rows = []
    t_cpu = 0.0
    for kk in range(1, args.fixed_steps + 1):
        sg = next(it_g)
        t0 = time.perf_counter()
        sc = next(it_c)
        t_cpu += time.perf_counter() - t0
        if kk in checkpoints:
            zg, zc = sg.z.numpy(), sc.z
            rows.append({"k": kk, "z_inf_diff_rel": float(np.max(np.abs(zg - zc)) / max(1.0, float(np.max(np.abs(zc))))),
                         "res_inf_gpu": float(sg.res_inf), "res_inf_cpu": float(np.max(np.abs(sc.res))),
                         "f_x_gpu": float(sg.f_x), "f_x_cpu": float(sc.f_x)})
    Fg, nzg = objective64(A, b, lam, sg.z.numpy())
    Fc, nzc = objective64(A, b, lam, sc.z)
    out["fixed"] = {"steps": args.fixed_steps, "checkpoints": rows, "objective_gpu": Fg, "objective_cpu": Fc,
                    "objective_rel_diff": abs(Fg - Fc) / abs(Fc), "nnz_gpu": nzg, "nnz_cpu": nzc,
                    "cpu_s_per_iteration": round(t_cpu / args.fixed_steps, 2)}

    # ---------------- adaptive step: gamma sequence / backtracking decisions (parity definition (ii)) ----------------
    if args.adaptive_steps > 0:
        it_g = iter(pa.FastForwardBackwardIteration(f=f_d, g=pa.NormL1(lam), x0=x0))
        it_c = iter(o.FastForwardBackwardIteration(f=o.LeastSquares(A, b), g=o.NormL1(lam), x0=x0))
        gam_g, gam_c = [], []
        for kk in range(1, args.adaptive_steps + 1):
            sg = next(it_g)
            sc = next(it_c)
            gam_g.append(float(sg.gamma))
            gam_c.append(float(sc.gamma))
        first_diff = next((i + 1 for i, (a, c) in enumerate(zip(gam_g, gam_c)) if abs(a - c) > 1e-6 * abs(c)), None)
        Fg, _ = objective64(A, b, lam, sg.z.numpy())
        Fc, _ = objective64(A, b, lam, sc.z)
        zg, zc = sg.z.numpy(), sc.z
        out["adaptive"] = {"steps": args.adaptive_steps, "gamma_gpu": gam_g, "gamma_cpu": gam_c,
                           "first_k_with_different_gamma": first_diff,
                           "z_inf_diff_rel": float(np.max(np.abs(zg - zc)) / max(1.0, float(np.max(np.abs(zc))))),
                           "objective_gpu": Fg, "objective_cpu": Fc, "objective_rel_diff": abs(Fg - Fc) / abs(Fc)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
