#!/usr/bin/env python3
"""Row teams on ONE GPU: the ranks of a row-sharded FastForwardBackward solve are contexts of this
process, one host thread and one stream each, every context limited to num_cu / ranks workgroups so that all team members
are resident together; the inboxes are plain device pointers (sharding.row_team_in_process), the registered collective
(initialisation, fallback) is a host-side double between the threads (tests/_doubles.ThreadAllReduce).  Every rank runs
the reference's iteration on its row block of A; the iterates are compared with the CPU restatement on the WHOLE matrix and
between the ranks (bit for bit).  Prints one JSON document.  Not two processes: their queues would alternate on one device
(profiles/r3_team_coop_vs_plain.md)."""
import argparse
import itertools
import json
import os
import sys
import threading

import numpy as np

# one HIP stream per rank, and every stream needs a hardware queue of its own: two sweeps that wait for each other on ONE
# queue would run one after the other (the runtime's default is four queues per process)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")  # (16 ranks + the default stream: with 16 queues two of the streams shared one and the self-test timed out)

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--fast", type=int, default=1)
    ap.add_argument("--adaptive", action="store_true", help="adaptive step (no Lf): FastForwardBackward's line search with the residual pair")
    ap.add_argument("--g", choices=["l1", "box", "l1w", "boxv"], default="l1", help="l1w / boxv: per-element weights / bounds (n-vectors, sliced under --cols)")
    ap.add_argument("--max-wgs", type=int, default=0, help="workgroups per rank (0: compute units / ranks)")
    ap.add_argument("--fault", type=int, default=0, help="rank 1's k-th row-team sweep loses a workgroup (0: none)")
    ap.add_argument("--fault-kind", type=int, default=0, help="0: that sweep loses a workgroup; 1: that sweep is refused at launch on rank 1")
    ap.add_argument("--cols", action="store_true",
                    help="COLUMN shards instead (rank p holds A[:, J_p] and the J_p slices of the n-vectors; one all-reduce of "
                         "m + 8 (N + 1) elements per iteration, the single sweep on every rank): same checks, slice by slice")
    ap.add_argument("--no-team", action="store_true", help="plain row shards (two sweeps + all-reduce) for comparison")
    ap.add_argument("--batched", action="store_true",
                    help="afterwards the same solve through the algorithm object with device_loop=True, check_every=4 (pg_iter_run_batched: "
                         "sweeps, finish kernels and scalar exchanges of four iterations enqueued back to back, one host read-back per batch)")
    ap.add_argument("--then-n", type=int, default=0,
                    help="afterwards a SECOND problem with this many columns on the same contexts (another ring layout: the "
                         "devices clear their inboxes and meet in one exchange before its first sweep)")
    ap.add_argument("--delay-ns", type=int, default=-1,
                    help="latency injector (pg_ctx_test_team_fault kind 2): the sweep's DELAY form, a step's granules accepted only "
                         "this many nanoseconds after they were stored (0: the injector's own cost; -1: off)")
    ap.add_argument("--tune", default="", help='run-time knobs of the row-team sweep for every rank, through pg_ctx_row_team_tune (no PG_TUNE): "PAIR=1,SPIN=4194304"')
    ap.add_argument("--solo", action="store_true",
                    help="--bench --ranks 1: a team of ONE rank (pg_ctx_test_team_fault kind 4) -- the sweep exchanges its granules with itself, the "
                         "kernel runs alone on the device: for rocprofv3 --pmc, which serialises kernels")
    ap.add_argument("--bench", action="store_true",
                    help="timing instead of parity: synthetic row blocks generated on the device (no host copy, no oracle), "
                         "--steps timed iterations after 3 warm-up steps; prints it/s and the aggregate bytes of A per second")
    args = ap.parse_args()
    if args.cols:
        args.no_team = True
    if args.bench:
        return bench(args)
    import proximalalgorithms.jl_amd as pa
    from _doubles import ThreadAllReduce
    from oracle import proxgrad_oracle as o
    from proximalalgorithms.jl_amd import _lib

    dtype = np.float32 if args.dtype == "f32" else np.float64
    m, n, N = args.m, args.n, args.ranks
    second = None
    if args.then_n:  # the follow-up problem, prepared up front (oracle iterates included)
        A2, b2, _ = o.synthetic_lasso(m, args.then_n, seed=4, dtype=dtype)
        lam2 = dtype(0.1) * dtype(np.max(np.abs(A2.T @ b2)))
        Lf2 = dtype(1.1) * dtype(np.linalg.norm(A2.astype(np.float64), 2) ** 2)
        ref2 = [s.z.copy() for s in itertools.islice(o.FastForwardBackwardIteration(f=o.LeastSquares(A2, b2), g=o.NormL1(lam2),
                                                                                   x0=np.zeros(args.then_n, dtype), Lf=Lf2), args.steps + 1)]
        second = (A2, b2, lam2, Lf2, ref2)
    A, b, _ = o.synthetic_lasso(m, n, seed=3, dtype=dtype)
    lam = dtype(0.1) * dtype(np.max(np.abs(A.T @ b)))
    v = np.ones(n, dtype) / dtype(np.sqrt(n))
    for _ in range(20):
        v = A.T @ (A @ v)
        v /= np.linalg.norm(v)
    Lf = dtype(1.1) * dtype(np.linalg.norm(A @ v) ** 2)
    if args.adaptive:
        Lf = None
    x0 = np.zeros(n, dtype)
    rv = np.random.default_rng(77)
    w_all = (lam * (0.25 + 1.5 * rv.random(n))).astype(dtype)
    lo_all = (-0.02 - 0.02 * rv.random(n)).astype(dtype)
    hi_all = (lo_all + dtype(0.05)).astype(dtype)
    mk_g = {"l1": lambda sl=slice(None): pa.NormL1(lam), "box": lambda sl=slice(None): pa.IndBox(dtype(-0.02), dtype(0.03)),
            "l1w": lambda sl=slice(None): pa.NormL1(w_all[sl].copy()), "boxv": lambda sl=slice(None): pa.IndBox(lo_all[sl].copy(), hi_all[sl].copy())}[args.g]
    mk_go = {"l1": lambda: o.NormL1(lam), "box": lambda: o.IndBox(dtype(-0.02), dtype(0.03)), "l1w": lambda: o.NormL1(w_all),
             "boxv": lambda: o.IndBox(lo_all, hi_all)}[args.g]
    Iter = pa.FastForwardBackwardIteration if args.fast else pa.ForwardBackwardIteration
    IterO = o.FastForwardBackwardIteration if args.fast else o.ForwardBackwardIteration
    ref_states = [(s.z.copy(), float(s.gamma)) for s in itertools.islice(IterO(f=o.LeastSquares(A, b), g=mk_go(), x0=x0, Lf=Lf), args.steps + 1)]
    ref = [r[0] for r in ref_states]

    comm = ThreadAllReduce(N)
    ctxs = [None] * N
    sync = threading.Barrier(N)
    results = [None] * N
    errors = []
    num_cu = pa.get_context().device_info()["compute_units"]
    max_wgs = args.max_wgs or max(1, num_cu // N)

    def worker(r):
        try:
            ctx = pa.Context.on_new_stream()
            ctxs[r] = ctx
            if args.cols:
                off, cnt = pa.shard_cols(n, N, r)
                A_loc = pa.HIPMatrix.from_numpy(np.asfortranarray(A[:, off:off + cnt]), ctx)
                f = pa.LeastSquares(A_loc, pa.HIPVector.from_numpy(b, ctx), comm=comm.view(r, "cols"))
                sl = slice(off, off + cnt)
            else:
                off, cnt = pa.shard_rows(m, N, r)
                A_loc = pa.HIPMatrix.from_numpy(np.asfortranarray(A[off:off + cnt]), ctx)
                f = pa.LeastSquares(A_loc, pa.HIPVector.from_numpy(b[off:off + cnt], ctx), comm=comm.view(r))
                sl = slice(None)
            sync.wait(timeout=120)
            if r == 0 and not args.no_team:
                pa.row_team_in_process(ctxs, max_wgs)
            if args.tune and not args.no_team:
                from proximalalgorithms.jl_amd.sharding import row_team_knobs_from_env

                pa.row_team_tune(ctx, **row_team_knobs_from_env(args.tune))
            sync.wait(timeout=120)
            selftest = None
            if not args.no_team:  # one scalar exchange through the inboxes: the ranks' contributions 1 + 2 + ... + N
                from proximalalgorithms.jl_amd.sharding import _row_team_selftest

                selftest = _row_team_selftest(ctx, N)
                sync.wait(timeout=120)
            if args.fault and r == 1:
                _lib.call("pg_ctx_test_team_fault", ctx.handle, args.fault, args.fault_kind)
            if args.delay_ns >= 0 and not args.no_team:
                _lib.call("pg_ctx_test_team_fault", ctx.handle, args.delay_ns, 2)
            iteration = Iter(f=f, g=mk_g(sl), x0=pa.HIPVector.from_numpy(x0[sl], ctx), Lf=Lf)
            rows, passes, zs = [], 0, []
            for k, s in enumerate(itertools.islice(iteration, args.steps + 1)):
                z = s.z.numpy()
                p = iteration.counters.get("a_passes", 0)
                rows.append({"k": k, "flags": int(getattr(s, "flags", 0)), "a_passes": int(p - passes), "f_x": float(s.f_x),
                             "gamma": float(s.gamma), "gamma_oracle": ref_states[k][1],
                             "dz": float(np.max(np.abs(z - ref[k][sl]))), "z_scale": float(max(1.0, np.max(np.abs(ref[k]))))})
                passes = p
                zs.append(z)
            batched = None
            if args.batched:
                zb, kb = pa.FastForwardBackward(tol=0.0, maxit=args.steps + 1, device_loop=True, check_every=4)(
                    x0=pa.HIPVector.from_numpy(x0[sl], ctx), f=f, g=mk_g(sl), Lf=Lf)
                zb = zb.numpy() if hasattr(zb, "numpy") else np.asarray(zb)
                batched = {"k": int(kb), "dz_rel": float(np.max(np.abs(zb - ref[args.steps][sl])) / max(1.0, float(np.max(np.abs(ref[args.steps])))))}
            second_dz = None
            if second is not None:
                A2, b2, lam2, Lf2, ref2 = second
                A2_loc = pa.HIPMatrix.from_numpy(np.asfortranarray(A2[off:off + cnt]), ctx)
                f2 = pa.LeastSquares(A2_loc, pa.HIPVector.from_numpy(b2[off:off + cnt], ctx), comm=comm.view(r))
                it2 = pa.FastForwardBackwardIteration(f=f2, g=pa.NormL1(lam2), x0=pa.HIPVector.zeros(args.then_n, dtype, ctx), Lf=Lf2)
                second_dz = []
                for k, s in enumerate(itertools.islice(it2, args.steps + 1)):
                    second_dz.append(float(np.max(np.abs(s.z.numpy() - ref2[k])) / max(1.0, float(np.max(np.abs(ref2[k]))))))
                second_dz = {"max_dz_rel": max(second_dz), "a_passes": int(it2.counters.get("a_passes", 0)),
                             "fallbacks": int(it2.counters.get("sweep_fallbacks", 0))}
            results[r] = (rows, zs, comm.calls[r], second_dz, batched, selftest, None if args.no_team else pa.row_team_geometry(ctx)[0])
        except BaseException as e:  # noqa: BLE001 -- reported in the JSON document, the other threads are released
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
    if args.cols:  # the ranks hold different slices; what they must agree on bit for bit are the iteration's scalars
        same = all(results[0][0][k][key] == results[r][0][k][key] for r in range(1, N) for k in range(args.steps + 1) for key in ("f_x", "gamma"))
    else:
        same = all(np.array_equal(results[0][1][k], results[r][1][k]) for r in range(1, N) for k in range(args.steps + 1))
    print(json.dumps({"m": m, "n": n, "ranks": N, "dtype": args.dtype, "max_wgs": max_wgs, "team": not args.no_team,
                      "ranks_agree_bitwise": bool(same), "fallback_flag": _lib.PG_FLAG_SWEEP_FALLBACK,
                      "allreduce_calls": [results[r][2] for r in range(N)], "steps": [results[r][0] for r in range(N)],
                      "second": [results[r][3] for r in range(N)], "batched": [results[r][4] for r in range(N)],
                      "selftest": [results[r][5] for r in range(N)], "geometry": [results[r][6] for r in range(N)]}))


def bench(args):
    """all ranks of a row team on ONE device, timed: what the PEER sweep streams when the members share the chip (the exchange
    then crosses no fabric -- a lower bound on its latency, an upper bound on nothing else)"""
    import math
    import time

    import proximalalgorithms.jl_amd as pa
    from _doubles import ThreadAllReduce

    dtype = np.float32 if args.dtype == "f32" else np.float64
    m, n, N = args.m, args.n, args.ranks
    comm = ThreadAllReduce(N)
    ctxs, out, errors = [None] * N, [None] * N, []
    sync = threading.Barrier(N)
    num_cu = pa.get_context().device_info()["compute_units"]
    max_wgs = args.max_wgs or max(1, num_cu // N)
    Lf = dtype(1.1 * (1.0 + math.sqrt(n / m)) ** 2)  # ||A||^2 of a Gaussian matrix scaled by 1 / sqrt(m), with margin

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
            if args.solo:
                from proximalalgorithms.jl_amd import _lib as _l

                _l.call("pg_ctx_test_team_fault", ctx.handle, 1, 4)
            if r == 0 and not args.no_team:
                pa.row_team_in_process(ctxs, max_wgs)
            if args.tune and not args.no_team:
                from proximalalgorithms.jl_amd.sharding import row_team_knobs_from_env

                pa.row_team_tune(ctx, **row_team_knobs_from_env(args.tune))
            sync.wait(timeout=300)
            if args.delay_ns >= 0 and not args.no_team:
                from proximalalgorithms.jl_amd import _lib

                _lib.call("pg_ctx_test_team_fault", ctx.handle, args.delay_ns, 2)
            iteration = pa.FastForwardBackwardIteration(f=f, g=pa.NormL1(dtype(0.05)), x0=pa.HIPVector.zeros(n, dtype, ctx), Lf=Lf)
            it = iter(iteration)
            for _ in range(4):
                s = next(it)
            p0 = iteration.counters.get("a_passes", 0)
            ctx.profile(True, kernels=("gemv_n_partial", "gemv_t", "gemv_tn"))
            ctx.profile_reset()
            ctx.sync()
            sync.wait(timeout=300)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                s = next(it)
            ctx.sync()
            sync.wait(timeout=300)
            dt = time.perf_counter() - t0
            prof = ctx.profile_read()
            slack = None
            if args.delay_ns >= 0 and not args.no_team:
                import ctypes as C

                tk, ws = C.c_int64(), C.c_int64()
                _lib.call("pg_ctx_test_team_slack", ctx.handle, C.byref(tk), C.byref(ws))
                slack = {"mean_us_between_store_and_use": (tk.value / max(ws.value, 1)) / 100.0, "wave_steps": ws.value}
            out[r] = {"seconds": dt, "slack": slack, "a_passes_per_step": (iteration.counters.get("a_passes", 0) - p0) / args.steps,
                      "row_team_stats": None if args.no_team else pa.row_team_stats(ctx),
                      "fallbacks": iteration.counters.get("sweep_fallbacks", 0), "res_inf": float(s.res_inf),
                      "kernels": {k: [v[0], round(v[1] / max(v[0], 1), 4)] for k, v in prof.items() if v[0]}}
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
    dt = max(o_["seconds"] for o_ in out)
    es = np.dtype(dtype).itemsize
    print(json.dumps({"bench": True, "m": m, "n": n, "ranks": N, "dtype": args.dtype, "team": not args.no_team, "max_wgs": max_wgs,
                      "delay_ns": args.delay_ns, "it_per_s": args.steps / dt, "ms_per_step": 1e3 * dt / args.steps,
                      "A_bytes_per_s_all_ranks": m * n * es * out[0]["a_passes_per_step"] * args.steps / dt, "ranks_out": out}))


if __name__ == "__main__":
    main()
