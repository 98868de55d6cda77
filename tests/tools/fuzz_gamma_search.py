#!/usr/bin/env python3
"""Randomised A/B of the step-size search of the PANOC family (fb_tools.jl:24-63; panoc.py::_backtrack_stepsize): three candidates of
gamma per read of A (pg_mat_mul_multi) against one product per candidate.  Per case a random problem -- rows 1 .. 9000 (below 13 row
groups the multi-vector product answers PG_ERR_UNSUPPORTED and the search must carry on with single products), 20 .. 1500 columns,
Float32 / Float64, logistic loss or squared distance, L1 / box, PANOC / ZeroFPR, a random scale of A (how often and how far gamma is
halved) and sometimes a minimum_gamma the search runs into -- is iterated twice on the GPU; required: gamma, tau and the iterate
BIT-identical at every iteration, the reads of A saved = the candidates taken from an earlier read, and (Float64) the oracle's
gamma.  Usage: python tests/tools/fuzz_gamma_search.py [cases] [first_seed]"""
import itertools
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def draw(seed):
    rng = np.random.default_rng(seed)
    f64 = bool(rng.random() < 0.5)
    m = int(rng.choice([1, 300, 1663, 1664, 1700, 3327, 3328, 3400, 4096, 5000, 8192, 9000])) if rng.random() < 0.6 else int(rng.integers(1, 9000))
    n = int(rng.integers(20, 1500))
    return dict(seed=seed, dtype=np.float64 if f64 else np.float32, m=m, n=n, loss=str(rng.choice(["logistic", "sqdist"])),
                g=str(rng.choice(["l1", "l1", "box"])), alg=str(rng.choice(["PANOCIteration", "ZeroFPRIteration"])),
                scale=float(10.0 ** rng.uniform(-0.5, 1.5)), min_gamma=(float(10.0 ** rng.uniform(-4, -1)) if rng.random() < 0.2 else 1e-7),
                its=int(rng.integers(4, 12)))


def one_case(seed):
    import proximalalgorithms.jl_amd as pa
    from oracle import proxgrad_oracle as o

    c = draw(seed)
    dtype = c["dtype"]
    label = "seed %d: %s %s %dx%d %s+%s scale %.3g min_gamma %.3g its %d" % (seed, c["alg"], dtype.__name__, c["m"], c["n"], c["loss"], c["g"], c["scale"],
                                                                           c["min_gamma"], c["its"])
    rng = np.random.default_rng(seed + 1)
    A = np.asfortranarray((c["scale"] * rng.standard_normal((c["m"], c["n"])) / np.sqrt(c["m"])).astype(dtype))
    b = (A @ (rng.standard_normal(c["n"]) * (rng.random(c["n"]) < 0.05)).astype(dtype) + dtype(0.01) * rng.standard_normal(c["m"]).astype(dtype)).astype(dtype)
    x0 = np.zeros(c["n"], dtype)
    fO = o.LogisticLoss(b) if c["loss"] == "logistic" else o.SquaredDistance(b)
    _, g0 = fO.value_and_gradient(np.zeros(c["m"], dtype))
    lam = dtype(0.1 * max(np.max(np.abs(A.T @ g0)), 1e-3))
    gD = pa.NormL1(lam) if c["g"] == "l1" else pa.IndBox(dtype(-0.3), dtype(0.5))
    gO = o.NormL1(lam) if c["g"] == "l1" else o.IndBox(dtype(-0.3), dtype(0.5))
    fD = pa.LogisticLoss(b) if c["loss"] == "logistic" else pa.SquaredDistance(b)
    M = pa.HIPMatrix.from_numpy(A)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        it3, it1 = (getattr(pa, c["alg"])(f=fD, A=M, g=gD, x0=x0, minimum_gamma=c["min_gamma"], gamma_candidates=k) for k in (3, 1))
        itO = getattr(o, c["alg"])(f=fO, A=A, g=gO, x0=x0, minimum_gamma=c["min_gamma"])
        sol = "xbar" if c["alg"] == "ZeroFPRIteration" else "z"
        for k, (s3, s1, so) in enumerate(itertools.islice(zip(it3, it1, itO), c["its"])):
            if float(s3.gamma) != float(s1.gamma) or float(s3.tau) != float(s1.tau):
                return "k=%d gamma / tau differ: %r %r | %r %r" % (k, float(s3.gamma), float(s1.gamma), float(s3.tau), float(s1.tau)), label, it3.counters
            if not np.array_equal(getattr(s3, sol).numpy(), getattr(s1, sol).numpy(), equal_nan=True):
                return "k=%d iterates differ" % k, label, it3.counters
            if dtype == np.float64 and k < 3 and np.isfinite(float(so.gamma)) and abs(float(s3.gamma) - float(so.gamma)) > 1e-9 * abs(float(so.gamma)):
                return "k=%d gamma %r, oracle %r" % (k, float(s3.gamma), float(so.gamma)), label, it3.counters
    taken = it3.counters.get("gamma_candidates_taken", 0)
    if it1.counters["A_passes"] - it3.counters["A_passes"] != taken:
        return "reads saved %d != candidates taken %d" % (it1.counters["A_passes"] - it3.counters["A_passes"], taken), label, it3.counters
    if it1.counters.get("gamma_candidates_ahead", 0) != 0:
        return "gamma_candidates = 1 evaluated ahead", label, it3.counters
    return None, label, it3.counters


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 31000
    t0 = time.time()
    bad, ahead, taken, multi = 0, 0, 0, 0
    for seed in range(first, first + cases):
        why, label, counters = one_case(seed)
        ahead += counters.get("gamma_candidates_ahead", 0)
        taken += counters.get("gamma_candidates_taken", 0)
        multi += 1 if counters.get("gamma_candidates_ahead", 0) else 0
        if why:
            bad += 1
            print("FAIL", label, why, flush=True)
    print("%d cases, %d failing, %d with candidates evaluated ahead (%d ahead, %d taken = reads of A saved), %.1f s" % (cases, bad, multi, ahead, taken, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
