#!/usr/bin/env python3
"""BASELINE config 1 and the reference's shipped instances: launch-bound sizes.  Reports solve time and it/s on the
GPU (host loop and in-library loop pg_iter_run) next to the CPU restatement -- these sizes are far below what a GPU is
for; the numbers document the per-iteration floor of the current (non-graph) path."""
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import proximalalgorithms.jl_amd as pa  # noqa: E402
from oracle import proxgrad_oracle as o  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def instances():
    A, b, _ = o.synthetic_lasso(200, 500, seed=0, dtype=np.float64)
    yield "config1 synthetic 200x500 f64", A, b, 0.1 * np.max(np.abs(A.T @ b))
    for name in ("lasso_tiny", "lasso_small", "lasso_medium"):
        d = np.load(os.path.join(GOLDEN, name + ".npz"))
        yield f"{name} {d['A'].shape[0]}x{d['A'].shape[1]} f64", d["A"], d["b"], float(d["lam"])


def main():
    pa.get_context()
    out = []
    for name, A, b, lam in instances():
        n = A.shape[1]
        x0 = np.zeros(n)
        f, g = pa.LeastSquares(A, b), pa.NormL1(lam)
        pa.FastForwardBackward(tol=1e-6)(x0=x0, f=f, g=g)  # warm-up
        t0 = time.perf_counter()
        z, k = pa.FastForwardBackward(tol=1e-6)(x0=x0, f=f, g=g)  # adaptive, like benchmark/benchmarks.jl:55-61
        t_host = time.perf_counter() - t0
        it = pa.FastForwardBackwardIteration(f=f, g=g, x0=x0)
        gen = iter(it)
        next(gen)
        t0 = time.perf_counter()
        k_lib, _ = it._fused.run(1, 10_000, 1e-6)
        t_lib = time.perf_counter() - t0
        its = pa.FastForwardBackwardIteration(f=f, g=g, x0=x0)
        next(iter(its))
        its._fused.run_small(1, 3, 1e-6)  # warm-up (code object load)
        its = pa.FastForwardBackwardIteration(f=f, g=g, x0=x0)
        next(iter(its))
        gc.collect()
        t0 = time.perf_counter()
        k_small, _ = its._fused.run_small(1, 10_000, 1e-6)
        t_small = time.perf_counter() - t0
        z_small = its._fused.view()["z"].numpy()
        # cooperative multi-workgroup solver: adaptive (the benchmark's mode) and fixed step, several grid sizes
        coop = {}
        for blocks in (0, 0, 8, 32, 64, 128, 256):  # the first pass is a warm-up
            itc = pa.FastForwardBackwardIteration(f=f, g=g, x0=x0)
            next(iter(itc))
            gc.collect()
            t0 = time.perf_counter()
            k_c, _ = itc._fused.run_coop(1, 10_000, 1e-6, blocks)
            t_c = time.perf_counter() - t0
            coop[blocks] = {"k": k_c, "it_s": k_c / t_c, "solve_ms": 1e3 * t_c}
        z_coop = itc._fused.view()["z"].numpy()
        Lf_c = float(np.linalg.norm(A, 2) ** 2)
        coop_fixed = {}
        for blocks in (0, 0, 8, 32, 64, 128, 256):
            itc = pa.FastForwardBackwardIteration(f=f, g=g, x0=x0, Lf=Lf_c)
            next(iter(itc))
            gc.collect()
            t0 = time.perf_counter()
            k_c, _ = itc._fused.run_coop(1, 4001, 0.0, blocks)
            coop_fixed[blocks] = (k_c - 1) / (time.perf_counter() - t0)
        from threadpoolctl import threadpool_limits

        t_cpu = None
        for threads in (1, 8):  # a BLAS pool sized for these matrices (64 threads are slower here); best of the two
            with threadpool_limits(limits=threads):
                o.fast_forward_backward(tol=1e-6, maxit=50, x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam))
                t0 = time.perf_counter()
                zo, ko = o.fast_forward_backward(tol=1e-6, x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam))
                t = time.perf_counter() - t0
            t_cpu = t if t_cpu is None else min(t_cpu, t)
        # fixed step (gamma = 1/Lf): one host synchronisation per iteration vs one per 32 iterations
        Lf = float(np.linalg.norm(A, 2) ** 2)
        fixed = {}
        for ce in (1, 1, 32):  # the first pass is a warm-up
            itf = pa.FastForwardBackwardIteration(f=f, g=g, x0=x0, Lf=Lf)
            next(iter(itf))
            gc.collect()  # destroying earlier iterators (hipFree) must not land inside the timed region
            t0 = time.perf_counter()
            kf, _ = itf._fused.run(1, 2001, 0.0, check_every=ce)
            fixed[ce] = (kf - 1) / (time.perf_counter() - t0)
        out.append({"instance": name, "gpu_fixed_step_it_s_sync_every_1": fixed[1], "gpu_fixed_step_it_s_sync_every_32": fixed[32], "k_gpu": k, "k_lib": k_lib, "k_cpu": ko, "gpu_host_loop_it_s": k / t_host,
                    "gpu_library_loop_it_s": k_lib / t_lib,
                    "k_persistent": k_small, "gpu_persistent_kernel_it_s": k_small / t_small,
                    "gpu_persistent_kernel_solve_ms": 1e3 * t_small, "cpu_numpy_solve_ms": 1e3 * t_cpu,
                    "gpu_coop_adaptive": {str(b_): v for b_, v in coop.items()},
                    "gpu_coop_fixed_step_it_s": {str(b_): v for b_, v in coop_fixed.items()},
                    "max_abs_diff_coop": float(np.max(np.abs(z_coop - zo))),
                    "max_abs_diff_persistent": float(np.max(np.abs(z_small - zo))), "cpu_numpy_it_s": ko / t_cpu,
                    "max_abs_diff": float(np.max(np.abs(z - zo)))})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
