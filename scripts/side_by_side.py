#!/usr/bin/env python3
"""N independent FastForwardBackward solves SIDE BY SIDE on one GPU, each on its own m x n LASSO (no exchange of any kind): what the
device streams when N kernels of N contexts (threads of one process, one stream each) or of N processes share it.  The calibration
for every "ranks sharing one device" figure of the row-team records: the same shapes, the plain single sweep, nothing to wait for.

    python scripts/side_by_side.py --m 2048 --n 1048576 --ranks 2 --mode threads|processes [--max-wgs-div K]
prints one JSON line: it/s per rank, bytes of A per second of all ranks together."""
import argparse
import json
import math
import os
import subprocess
import sys
import threading
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def solve(args, r, ready=None, go=None):
    import proximalalgorithms.jl_amd as pa

    dtype = np.float32
    ctx = pa.Context.on_new_stream() if args.mode == "threads" else pa.get_context(0)
    A = pa.HIPMatrix.synthetic(args.m, args.n, dtype, seed=r, ctx=ctx)
    xt = np.zeros(args.n, dtype)
    xt[np.random.default_rng(5).choice(args.n, size=max(1, args.n // 1000), replace=False)] = 1.0
    b = A.mul(pa.HIPVector.from_numpy(xt, ctx))
    Lf = dtype(1.1 * (1.0 + math.sqrt(args.n / args.m)) ** 2)
    it = iter(pa.FastForwardBackwardIteration(f=pa.LeastSquares(A, b), g=pa.NormL1(dtype(0.05)), x0=pa.HIPVector.zeros(args.n, dtype, ctx), Lf=Lf))
    for _ in range(4):
        next(it)
    ctx.sync()
    if ready is not None:
        ready()
    if go is not None:
        go()
    t0 = time.time()
    for _ in range(args.steps):
        next(it)
    ctx.sync()
    t1 = time.time()
    return {"rank": r, "t0": t0, "t1": t1, "it_per_s": args.steps / (t1 - t0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=2048)
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--mode", choices=["threads", "processes"], default="threads")
    ap.add_argument("--child", type=int, default=-1)
    ap.add_argument("--start-at", type=float, default=0.0)
    args = ap.parse_args()
    if args.child >= 0:  # one process of --mode processes: start the timed part at a wall-clock instant all children share
        def go():
            while time.time() < args.start_at:
                time.sleep(0.0005)
        print(json.dumps(solve(args, args.child, go=go)), flush=True)
        return
    if args.mode == "threads":
        out, bar = [None] * args.ranks, threading.Barrier(args.ranks)
        def worker(r):
            out[r] = solve(args, r, go=lambda: bar.wait(timeout=600))
        ts = [threading.Thread(target=worker, args=(r,)) for r in range(args.ranks)]
        [t.start() for t in ts]
        [t.join() for t in ts]
    else:
        start_at = time.time() + 60.0  # (a fresh box imports torch for a minute; children that are late start at once and the overlap shows it)
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--m", str(args.m), "--n", str(args.n), "--steps", str(args.steps),
                                "--mode", "processes", "--child", str(r), "--start-at", repr(start_at)], stdout=subprocess.PIPE, text=True)
              for r in range(args.ranks)]
        out = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in ps]
    t0, t1 = max(o["t0"] for o in out), min(o["t1"] for o in out)
    span = max(o["t1"] for o in out) - min(o["t0"] for o in out)
    total = sum(o["it_per_s"] for o in out)
    print(json.dumps({"side_by_side": args.mode, "m": args.m, "n": args.n, "ranks": args.ranks, "steps": args.steps,
                      "it_per_s": [round(o["it_per_s"], 2) for o in out], "overlap_fraction": round(max(0.0, t1 - t0) / span, 3),
                      "TBps_all_ranks": round(total * args.m * args.n * 4 / 1e12, 3)}))


if __name__ == "__main__":
    main()
