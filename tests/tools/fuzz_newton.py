#!/usr/bin/env python3
"""Randomised differential test of the widened rows: PANOC / ZeroFPR / PANOCplus (L-BFGS directions,
general A, squared-distance and logistic losses) and DouglasRachford (separable quadratic + box / L1, stepping and the
K-iterations-per-sweep loop) against the CPU restatement.  Usage: python tests/tools/fuzz_newton.py [cases] [first_seed] [tall].
`tall`: PANOC / ZeroFPR / PANOCplus only, column lengths 600 .. 140000 with few columns.  `wide`: ZeroFPR / PANOCplus on under-determined
problems whose columns are 33 .. 64 row groups long (the two- and three-point sweeps' range; the line searches backtrack there)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import proximalalgorithms.jl_amd as pa  # noqa: E402
from oracle import proxgrad_oracle as o  # noqa: E402


TALL = False
WIDE = False


def one_case(seed):
    rng = np.random.default_rng(seed)
    dtype = np.float64 if rng.random() < 0.7 else np.float32
    alg = rng.choice(["panoc", "zerofpr", "panocplus", "dr"])
    fails = []
    if alg == "dr" and not TALL and not WIDE:
        n = int(rng.choice([1, 7, 64, 1000, 4099])) if rng.random() < 0.5 else int(rng.integers(1, 20000))
        vec = rng.random() < 0.7
        d = (0.1 + np.abs(rng.standard_normal(n))).astype(dtype) if vec else dtype(0.1 + rng.random())
        q = rng.standard_normal(n).astype(dtype) if (vec or rng.random() < 0.5) else dtype(rng.standard_normal())
        gamma = dtype(0.2 + 2 * rng.random())
        x0 = rng.standard_normal(n).astype(dtype)
        box = rng.random() < 0.6
        g_g, g_o = (pa.IndBox(dtype(-0.4), dtype(0.3)), o.IndBox(dtype(-0.4), dtype(0.3))) if box else (pa.NormL1(dtype(0.15)), o.NormL1(dtype(0.15)))
        tol = 1e-4 if dtype == np.float32 else float(rng.choice([1e-6, 1e-10]))
        maxit = int(rng.choice([7, 40, 500]))
        y_o, k_o = o.douglas_rachford(tol=tol, maxit=maxit, x0=x0, f=o.SeparableQuadratic(d, q), g=g_o, gamma=gamma)
        for loop, blk in (("host", 1), ("device", 8), ("device", 16), ("device", 64), ("device", 1)):
            y, k = pa.DouglasRachford(tol=tol, maxit=maxit, device_loop=(loop == "device"), check_every=blk)(
                x0=x0, f=pa.SeparableQuadratic(d, q), g=g_g, gamma=gamma)
            if k != k_o or not np.array_equal(y, y_o):  # the separable prox is evaluated without contraction: same bits
                fails.append((loop, blk, f"k={k} k_cpu={k_o} dy={np.max(np.abs(y - y_o)) if n else 0:.2e}"))
        return f"seed={seed} DR {np.dtype(dtype).name} n={n} {'box' if box else 'l1'} tol={tol} maxit={maxit} k_cpu={k_o}", fails
    m, n = int(rng.integers(2, 300)), int(rng.integers(2, 500))
    if TALL:  # every geometry of the single sweep (also in its A (x - z) mode, the L-BFGS image slab's feed): column lengths up to the teams'
        m = int(rng.choice([600, 2048, 2305, 4096, 5000, 7168, 9000, 16384, 20000, 33000, 40000, 70001, 131072]))
        if rng.random() < 0.4:
            m = int(rng.integers(2049, 140000))
        n = int(rng.choice([2, 7, 33, 64, 130]))
        alg = str(rng.choice(["panoc", "zerofpr", "panocplus"]))
    if WIDE:  # columns of 33 .. 64 row groups of 1 KiB, more columns than rows, a sparse solution: ZeroFPR's search goes to tau / 4
        m = int(rng.integers(33 * 1024 // np.dtype(dtype).itemsize - 200, 64 * 1024 // np.dtype(dtype).itemsize + 1))
        n = int(m * rng.uniform(1.2, 2.0))
        alg = str(rng.choice(["zerofpr", "zerofpr", "panocplus"]))
    A = np.asfortranarray(rng.standard_normal((m, n)).astype(dtype) / dtype(np.sqrt(m)))
    xt = np.zeros(n, dtype)
    nzc = max(1, n // (1000 if WIDE else 10))
    xt[rng.choice(n, nzc, replace=False)] = rng.standard_normal(nzc).astype(dtype)
    b = (A @ xt + dtype(0.01) * rng.standard_normal(m).astype(dtype)).astype(dtype)
    loss = rng.choice(["sqdist", "logistic"])
    L, Lo = (pa.SquaredDistance, o.SquaredDistance) if loss == "sqdist" else (pa.LogisticLoss, o.LogisticLoss)
    lam = dtype(0.1) * dtype(np.max(np.abs(A.T @ b))) if loss == "sqdist" else dtype(0.02)
    box = rng.random() < 0.25
    g_g, g_o = (pa.IndBox(dtype(-0.5), dtype(0.5)), o.IndBox(dtype(-0.5), dtype(0.5))) if box else (pa.NormL1(lam), o.NormL1(lam))
    tol = 1e-4 if dtype == np.float32 else 1e-7
    x0 = np.zeros(n, dtype)
    G, O = {"panoc": (pa.PANOC, o.panoc), "zerofpr": (pa.ZeroFPR, o.zerofpr), "panocplus": (pa.PANOCplus, o.panocplus)}[alg]
    z_o, k_o = O(tol=tol, maxit=400, x0=x0, f=Lo(b), A=A, g=g_o)
    z, k = G(tol=tol, maxit=400)(x0=x0, f=L(b), A=A, g=g_g)
    A64, b64 = A.astype(np.float64), b.astype(np.float64)

    def obj(v):
        t = A64 @ v.astype(np.float64) - b64
        fv = 0.5 * np.sum(t * t) if loss == "sqdist" else np.sum(np.log1p(np.exp(-t)))
        return fv + (0.0 if box else float(lam) * np.sum(np.abs(v)))

    equalized = (k_o >= 400) != (k >= 400)
    if equalized:
        # one side stopped on its rule, the other ran into maxit: two different points of a still descending objective.  Compare
        # at EQUAL iteration counts instead of widening the tolerance: re-run the longer side for the shorter count.
        kk = min(k, k_o)
        if k > kk:
            z, k = G(tol=0.0, maxit=kk)(x0=x0, f=L(b), A=A, g=g_g)
        else:
            z_o, k_o = O(tol=0.0, maxit=kk, x0=x0, f=Lo(b), A=A, g=g_o)
    F, F_o = obj(z), obj(z_o)
    dF = abs(F - F_o) / max(abs(F_o), 1e-3 * obj(x0))
    # runs the CPU did not converge either (logistic loss on separable data, m << n) are compared loosely: the
    # quasi-Newton trajectories amplify rounding and only share the objective level
    # f32: the stop rule (1e-4 on res / gamma) bounds a flat objective only this far.  (One side at maxit used to get twice the
    # allowance -- seed 144: F = 1.39e-3 at k = 236 on the device, 1.21e-3 at k = 400 on the CPU; such pairs are now compared
    # at equal iteration counts, above.)
    # (both sides at maxit: neither converged -- the quasi-Newton trajectories only share the objective level)
    Ftol = 1e-2 if equalized else (2e-2 if (k_o >= 400 and k >= 400) else (1e-2 if dtype == np.float32 else 1e-6))
    if dF > Ftol or (k_o < 400 and k > max(k_o + 15, 2 * k_o)):
        fails.append((alg, f"k={k} k_cpu={k_o} dF={dF:.2e}"))
    return f"seed={seed} {alg} {np.dtype(dtype).name} {m}x{n} {loss} {'box' if box else 'l1'} k_cpu={k_o}", fails


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    global TALL, WIDE
    TALL = "tall" in sys.argv[3:]
    WIDE = "wide" in sys.argv[3:]
    pa.get_context()
    bad, t0 = 0, time.perf_counter()
    for seed in range(first, first + cases):
        desc, fails = one_case(seed)
        if fails:
            bad += 1
            print("FAIL", desc, fails, flush=True)
    print(f"{cases} cases, {bad} failing, {time.perf_counter() - t0:.1f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
