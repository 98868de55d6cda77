#!/usr/bin/env python3
"""The reference's own benchmark suite (benchmark/benchmarks.jl:30-135) on the device: for each shipped instance
(lasso tiny 5x10, small 50x100, medium 500x1000, Float64) the eleven solver calls of the suite, timed like
BenchmarkTools does (operators built in an untimed setup before every repetition, the solver call timed; median of
``--repeat`` runs), next to the CPU
restatement (oracle/, numpy) running the same call on this box's host cores.  Prints one JSON line per (instance,
solver): iterations, device ms, CPU ms, objective of both answers.

These sizes are launch-bound (A is 400 B .. 4 MB): the device numbers document the per-iteration floor of the
host-stepped path and, for ForwardBackward / FastForwardBackward, of the in-library loops (``device_loop=True``).
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import proximalalgorithms.jl_amd as pa  # noqa: E402
from oracle import proxgrad_oracle as o  # noqa: E402
from oracle import proxgrad_oracle_ext as ox  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def suite(A, b, lam):
    """(name, device call, CPU call) for the solver entries of benchmarks.jl:47-134"""
    n, m = A.shape[1], A.shape[0]
    T = A.dtype
    opn2 = float(np.linalg.norm(A, 2) ** 2)
    Lf_drls = float(np.linalg.norm(A.T @ A, 2))  # benchmarks.jl:103
    x0 = np.zeros(n, T)

    def dev(solver, kw):
        # (setup, call) like BenchmarkTools' `setup` / timed expression: the operators (device upload of A, b) are built
        # outside the timed region, every repetition gets fresh ones (a cached prox factorisation is then rebuilt inside
        # the timed call, as the reference's LeastSquares does on its first prox!)
        return (kw, lambda built: solver(x0=x0, **built))

    ls = lambda: dict(f=pa.LeastSquares(A, b), g=pa.NormL1(lam))
    sq = lambda: dict(f=pa.SquaredDistance(b), A=A, g=pa.NormL1(lam))
    ols = lambda: dict(f=o.LeastSquares(A, b), g=o.NormL1(lam))
    osq = lambda: dict(f=o.SquaredDistance(b), A=A, g=o.NormL1(lam))
    rows = [
        ("ForwardBackward", pa.ForwardBackward(tol=1e-6), ls, lambda: o.forward_backward(tol=1e-6, x0=x0, **ols())),
        ("ForwardBackward[device_loop]", pa.ForwardBackward(tol=1e-6, device_loop=True), ls, None),
        ("FastForwardBackward", pa.FastForwardBackward(tol=1e-6), ls, lambda: o.fast_forward_backward(tol=1e-6, x0=x0, **ols())),
        ("FastForwardBackward[device_loop]", pa.FastForwardBackward(tol=1e-6, device_loop=True), ls, None),
        ("ZeroFPR", pa.ZeroFPR(tol=1e-6), sq, lambda: o.zerofpr(tol=1e-6, x0=x0, **osq())),
        ("PANOC", pa.PANOC(tol=1e-6), sq, lambda: o.panoc(tol=1e-6, x0=x0, **osq())),
        ("PANOCplus", pa.PANOCplus(tol=1e-6), sq, lambda: o.panocplus(tol=1e-6, x0=x0, **osq())),
        ("DouglasRachford", pa.DouglasRachford(tol=1e-6), lambda: dict(gamma=1.0, **ls()),
         lambda: o.douglas_rachford(tol=1e-6, x0=x0, gamma=1.0, **ols())),
        ("DRLS", pa.DRLS(tol=1e-6), lambda: dict(Lf=Lf_drls, **ls()), lambda: ox.drls(tol=1e-6, x0=x0, Lf=Lf_drls, **ols())),
        ("AFBA-1", pa.AFBA(theta=1, mu=1, tol=1e-6), lambda: dict(y0=np.zeros(n, T), beta_f=opn2, **ls()),
         lambda: ox.afba(theta=1, mu=1, tol=1e-6, x0=x0, y0=np.zeros(n, T), beta_f=opn2, **ols())),
        ("AFBA-2", pa.AFBA(theta=1, mu=1, tol=1e-6),
         lambda: dict(y0=np.zeros(m, T), h=pa.SquaredDistance(b), L=A, g=pa.NormL1(lam)),
         lambda: ox.afba(theta=1, mu=1, tol=1e-6, x0=x0, y0=np.zeros(m, T), h=ox.SqrDistance(b), L=A, g=o.NormL1(lam))),
        ("SFISTA", pa.SFISTA(tol=1e-3), lambda: dict(Lf=opn2, **ls()), lambda: ox.sfista(tol=1e-3, x0=x0, Lf=opn2, **ols())),
        # the same calls with the iteration body replayed as a hipGraph (one launch per iteration + the stop test)
        ("DouglasRachford[graph]", pa.DouglasRachford(tol=1e-6, graph=True), lambda: dict(gamma=1.0, **ls()), None),
        ("AFBA-1[graph]", pa.AFBA(theta=1, mu=1, tol=1e-6, graph=True), lambda: dict(y0=np.zeros(n, T), beta_f=opn2, **ls()), None),
        ("AFBA-2[graph]", pa.AFBA(theta=1, mu=1, tol=1e-6, graph=True),
         lambda: dict(y0=np.zeros(m, T), h=pa.SquaredDistance(b), L=A, g=pa.NormL1(lam)), None),
    ]
    return [(name, dev(solver, kw), cpu) for name, solver, kw, cpu in rows]


def objective(A, b, lam, x):
    x = x[0] if isinstance(x, tuple) else x
    return float(0.5 * np.sum((A @ x - b) ** 2) + lam * np.sum(np.abs(x)))


def timed(call, repeat):
    """call: a zero-argument function, or a (setup, fn) pair -- setup() runs untimed before every repetition"""
    setup, fn = call if isinstance(call, tuple) else (None, call)
    ts, out = [], None
    for _ in range(repeat):
        built = setup() if setup is not None else None
        t0 = time.perf_counter()
        out = fn(built) if setup is not None else fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts) * 1e3, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--instances", default="tiny,small,medium")
    ap.add_argument("--only", default="", help="substring filter on the solver name")
    ap.add_argument("--cpu-threads", type=int, default=8, help="BLAS threads for the CPU restatement's runs")
    args = ap.parse_args()
    pa.set_default_context(pa.Context.on_new_stream())  # a capturable (non-default) stream for the [graph] rows
    for inst in args.instances.split(","):
        d = np.load(os.path.join(GOLDEN, f"lasso_{inst}.npz"))
        A, b, lam = np.asfortranarray(d["A"].astype(np.float64)), d["b"].astype(np.float64), float(d["lam"])
        f_star = objective(A, b, lam, d["xstar"])
        for name, dev_call, cpu_call in suite(A, b, lam):
            if args.only and args.only not in name:
                continue
            dev_call[1](dev_call[0]())  # warm-up (code objects, workspaces)
            ms_dev, (xd, it_dev) = timed(dev_call, args.repeat)
            rec = {"instance": f"lasso_{inst} {A.shape[0]}x{A.shape[1]} f64", "solver": name, "iterations": int(it_dev),
                   "device_ms": round(ms_dev, 3), "device_us_per_iteration": round(1e3 * ms_dev / max(it_dev, 1), 2),
                   "objective_gap_device": objective(A, b, lam, xd) - f_star}
            if cpu_call is not None:
                # the CPU restatement with a BLAS pool sized for these matrices (64 threads on a 500 x 1000 GEMV are
                # slower than 8: the pool's wake-ups dominate)
                from threadpoolctl import threadpool_limits

                with threadpool_limits(limits=args.cpu_threads):
                    ms_cpu, (xc, it_cpu) = timed(cpu_call, max(1, args.repeat - 1))
                rec["cpu_threads"] = args.cpu_threads
                rec.update(cpu_iterations=int(it_cpu), cpu_ms=round(ms_cpu, 3),
                           objective_gap_cpu=objective(A, b, lam, xc) - f_star)
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
