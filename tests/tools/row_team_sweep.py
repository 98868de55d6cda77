#!/usr/bin/env python3
"""Row-team sweep on ONE GPU under an injected hand-off latency.

The ranks of a row team are contexts of this process (as in tests/tools/row_team.py --bench); the row blocks are generated
once, then every (geometry, injected latency) pair of the plan is timed on them: the geometry through the tuning variables
(PG_TUNE=1: PG_TNP_C / _LAG / _LAGR / _PF / _WGS / _W, read by the library at every launch), the latency through
pg_ctx_test_team_fault(ctx, ns, 2) -- the sweep's DELAY form accepts a step's granules only `ns` after they were stored, so
the on-chip hand-off stands in for a fabric hop of that length (pg_gemv_tnt.h).  One JSON line per pair.

    python tests/tools/row_team_sweep.py --m 4096 --n 1048576 --geoms default,2:2:3:2:2:3 --delays off,0,2000,4000,8000
geometry = C:LAG:LAGR:PF:WGS:W[:K1] (columns per step, lag steps in LDS, lag steps in registers, tiles in flight, workgroups per CU, waves per
workgroup; K1 = 0: round 5's kernel where round 6's one-wave sweep is the default)."""
import argparse
import ctypes as C
import json
import math
import os
import sys
import threading
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
os.environ["PG_TUNE"] = "1"

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GEOM_VARS = ("PG_TNP_C", "PG_TNP_LAG", "PG_TNP_LAGR", "PG_TNP_PF", "PG_TNP_WGS", "PG_TNP_W", "PG_TNP_K1", "PG_TNP_PAIR", "PG_TNP_AHEAD")


def set_geometry(spec):
    for v in GEOM_VARS:
        os.environ.pop(v, None)
    if spec != "default":
        for v, x in zip(GEOM_VARS, spec.split(":")):
            os.environ[v] = x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--geoms", default="default")
    ap.add_argument("--delays", default="off,0", help="comma list of nanoseconds; 'off' = the product kernel (no injector)")
    ap.add_argument("--two-sweeps", action="store_true", help="also time the two-sweep row layout on the same blocks (first line)")
    args = ap.parse_args()

    import proximalalgorithms.jl_amd as pa
    from _doubles import ThreadAllReduce
    from proximalalgorithms.jl_amd import _lib

    dtype = np.float32 if args.dtype == "f32" else np.float64
    m, n, N = args.m, args.n, args.ranks
    plan = [(g, d) for _ in range(args.repeat) for g in args.geoms.split(",") for d in args.delays.split(",")]
    comm = ThreadAllReduce(N)
    ctxs, errors = [None] * N, []
    sync = threading.Barrier(N)
    Lf = dtype(1.1 * (1.0 + math.sqrt(n / m)) ** 2)
    es = np.dtype(dtype).itemsize
    lines = []

    def stats(ctx):
        s = pa.row_team_stats(ctx)
        tk, ws = C.c_int64(), C.c_int64()
        _lib.call("pg_ctx_test_team_slack", ctx.handle, C.byref(tk), C.byref(ws))
        return s["late_waves"], s["wait_polls"], tk.value, ws.value

    def timed(r, ctx, f, label, delay, team):
        iteration = pa.FastForwardBackwardIteration(f=f, g=pa.NormL1(dtype(0.05)), x0=pa.HIPVector.zeros(n, dtype, ctx), Lf=Lf)
        it = iter(iteration)
        for _ in range(4):
            s = next(it)
        p0 = iteration.counters.get("a_passes", 0)
        fb0 = iteration.counters.get("sweep_fallbacks", 0)
        st0 = stats(ctx) if team else (0, 0, 0, 0)
        ctx.sync()
        sync.wait(timeout=300)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            s = next(it)
        ctx.sync()
        sync.wait(timeout=300)
        dt = time.perf_counter() - t0
        st1 = stats(ctx) if team else (0, 0, 0, 0)
        passes = (iteration.counters.get("a_passes", 0) - p0) / args.steps
        out = {"seconds": dt, "a_passes_per_step": passes, "fallbacks": iteration.counters.get("sweep_fallbacks", 0) - fb0,
               "late_waves": st1[0] - st0[0], "wait_polls": st1[1] - st0[1],
               "slack_us": ((st1[2] - st0[2]) / max(st1[3] - st0[3], 1)) / 100.0 if st1[3] > st0[3] else None,
               "res_inf": float(s.res_inf)}
        results[r] = out
        sync.wait(timeout=300)
        if r == 0:
            dtm = max(o_["seconds"] for o_ in results)
            line = {"m": m, "n": n, "ranks": N, "dtype": args.dtype, "geometry": label, "delay_ns": delay, "steps": args.steps,
                    "it_per_s": args.steps / dtm, "ms_per_step": 1e3 * dtm / args.steps,
                    "TBps_all_ranks": m * n * es * results[0]["a_passes_per_step"] / (dtm / args.steps) / 1e12,
                    "a_passes_per_step": results[0]["a_passes_per_step"], "fallbacks": [o_["fallbacks"] for o_ in results],
                    "late_waves": [o_["late_waves"] for o_ in results], "wait_polls": [o_["wait_polls"] for o_ in results],
                    "slack_us": [o_["slack_us"] for o_ in results], "res_inf": [o_["res_inf"] for o_ in results]}
            lines.append(line)
            print(json.dumps(line), flush=True)
        sync.wait(timeout=300)

    results = [None] * N

    def worker(r):
        try:
            ctx = pa.Context.on_new_stream()
            ctxs[r] = ctx
            off, cnt = pa.shard_rows(m, N, r)
            A_loc = pa.HIPMatrix.synthetic(cnt, n, dtype, seed=0, row_offset=off, m_global=m, ctx=ctx)
            rng = np.random.default_rng(5)
            xt = np.zeros(n, dtype)
            xt[rng.choice(n, size=max(1, n // 1000), replace=False)] = 1.0
            b = A_loc.mul(pa.HIPVector.from_numpy(xt, ctx))
            f = pa.LeastSquares(A_loc, b, comm=comm.view(r))
            sync.wait(timeout=300)
            if args.two_sweeps:
                timed(r, ctx, f, "two-sweeps", None, False)
            if r == 0:
                pa.row_team_in_process(ctxs, -N)
            sync.wait(timeout=300)
            for geom, delay in plan:
                if r == 0:
                    set_geometry(geom)
                sync.wait(timeout=300)
                if delay == "off":
                    _lib.call("pg_ctx_test_team_fault", ctx.handle, 0, 3)
                else:
                    _lib.call("pg_ctx_test_team_fault", ctx.handle, int(delay), 2)
                try:
                    timed(r, ctx, f, geom, None if delay == "off" else int(delay), True)
                except pa.ProxGradError as e:  # an instantiation that does not exist: every rank gets the same refusal
                    if r == 0:
                        print(json.dumps({"geometry": geom, "delay_ns": delay, "error": str(e)[:200]}), flush=True)
                    sync.wait(timeout=300)
        except BaseException as e:  # noqa: BLE001
            import traceback

            errors.append("rank %d: %s\n%s" % (r, e, traceback.format_exc()))
            sync.abort()
            comm.barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(N)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        print(json.dumps({"error": errors}))
        sys.exit(1)


if __name__ == "__main__":
    main()
