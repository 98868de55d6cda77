#!/usr/bin/env python3
"""bench.py -- FastForwardBackward iterations/sec on synthetic LASSO (BASELINE.json metric) on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is ONE FastForwardBackward iteration (fast_forward_backward.jl:106-145) of the fused HIP engine on
the headline workload  m = 16384, n = 2^20, Float32, fixed step gamma = 1/Lf  (BASELINE.json north_star).
A (64 GiB) is generated on the device and is resident in HBM before the timed region.  By default an iteration
reads A ONCE (the single-sweep iteration: A' r, prox, next extrapolation and next residual per column while
it is in registers); --sweeps two runs A x and A' r as separate sweeps like the reference.  For N > 1 the
driver launches one process per GPU with torch.distributed.run (STRONG scaling: the global problem is fixed):
column blocks of A are sharded over the N ranks, which keeps the single sweep on every GPU with ONE RCCL
all-reduce of m + 4 N floats per iteration (--sharding rows: row blocks, two sweeps, n+1 floats per gradient
evaluation -- north_star's layout, used for the adaptive mode and weak scaling).  Rank 0 prints ONE JSON line.

Extra legs in the same line:
  roofline      HBM roofline of the dominant kernel (the slowest sweep over A), timed live with HIP event pairs
                on the launch stream (pg_ctx_profile_*), algorithmic bytes = one full read of the local A block +
                its vectors; whole_iteration reports the SURVEY 8(d) two-pass figure and the bytes actually moved.
  cpu_baseline  the CPU restatement (oracle/, numpy + OpenBLAS, same unfused op order as the reference) timed on
                this host on the SAME workload (the device matrix copied to host memory) when memory allows, else
                on a bounded column sample scaled to it/s of the full workload.
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (m, n)            BASELINE.json
    "headline": (16384, 1 << 20),  # north_star target; configs[4] is its 8-GPU weak-scaled twin
    "config2": (8192, 262144),  # configs[1]
    "small": (2048, 16384),  # quick functional check
}


def parse_args():
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--workload", choices=sorted(WORKLOADS), default="headline")
    p.add_argument("--m", type=int, default=None, help="override rows (global)")
    p.add_argument("--n", type=int, default=None, help="override columns")
    p.add_argument("--mode", choices=["fixed", "adaptive"], default="fixed")
    p.add_argument("--dtype", choices=["f32", "f64"], default="f32", help="working precision (BASELINE metric: f32)")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--sweeps", choices=["one", "two"], default="one",
                   help="one: the single-sweep iteration (A read once per iteration; single GPU) -- two: A x and A' r as "
                        "separate sweeps like the reference (always the case when rows are sharded)")
    p.add_argument("--sharding", choices=["auto", "rows", "cols"], default="auto",
                   help="N > 1: how A is distributed (auto: column blocks for the fixed-step single-sweep run, row blocks "
                        "otherwise)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--kernel-events", choices=["gemv", "all", "none"], default="gemv",
                   help="which kernels are bracketed by HIP event pairs in the timed region (none: no roofline object)")
    p.add_argument("--cpu-baseline", choices=["auto", "full", "sample"], default="auto",
                   help="CPU leg on the downloaded full matrix (when host memory allows) or on a column sample")
    p.add_argument("--cpu-sample-cols", type=int, default=16384)
    p.add_argument("--cpu-steps", type=int, default=10)
    p.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                   help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for functional tests)")
    p.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                   help="strong: the global problem is fixed and its rows are split over the ranks (default); weak: every "
                        "rank holds --m rows (BASELINE config 5 = --scaling weak --m 16384 on 8 GPUs: 131072 x 2^20)")
    p.add_argument("--collective", choices=["torch", "native"], default="torch",
                   help="N > 1: all-reduce through torch.distributed (default) or the library's own RCCL communicator")
    p.add_argument("--overlap", action="store_true",
                   help="pipeline the [grad ; f] all-reduce with pass T in column chunks (N > 1). Off by default: at the "
                        "headline shard shape the chunking costs ~60 us/step, about what it can hide (DESIGN.md section 6)")
    p.add_argument("--no-overlap", action="store_true", help="(default; kept for older command lines)")
    p.add_argument("--force-comm", action="store_true",
                   help="diagnostic: attach the collective even with one rank (measures the cost of the sharded code path)")
    p.add_argument("--share-device", action="store_true",
                   help="functional test mode: every rank uses cuda:0 (e.g. 2 ranks on a 1-GPU box, with --backend gloo)")
    return p.parse_args()


def _host_memory_limit():
    """Bytes this process may use: the cgroup limit when there is one, else MemAvailable."""
    lim = None
    try:
        v = open("/sys/fs/cgroup/memory.max").read().strip()
        if v != "max":
            lim = int(v)
    except Exception:
        pass
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                avail = int(ln.split()[1]) * 1024
                lim = avail if lim is None else min(lim, avail)
    except Exception:
        pass
    return lim


def cpu_baseline_full(A_dev, b_dev, lam, Lf, budget_s=25.0, max_steps=8):
    """The SAME workload on the host cores: the device matrix is copied to host memory (a few seconds over PCIe) and the
    oracle (numpy/OpenBLAS restatement of the reference's op sequence) steps on it for at most `budget_s` seconds."""
    import numpy as np

    from oracle import proxgrad_oracle as o

    t0 = time.perf_counter()
    A = A_dev.numpy()
    b = b_dev.numpy()
    t_dl = time.perf_counter() - t0
    m, n = A.shape
    it = iter(o.FastForwardBackwardIteration(f=o.LeastSquares(A, b), g=o.NormL1(lam), x0=np.zeros(n, A.dtype), Lf=Lf))
    next(it)  # init (two passes), untimed like the GPU side
    steps, t0 = 0, time.perf_counter()
    while steps < max_steps and (steps < 2 or time.perf_counter() - t0 < budget_s):
        next(it)
        steps += 1
    dt = time.perf_counter() - t0
    one_thread = None
    try:
        from threadpoolctl import threadpool_info, threadpool_limits

        cores = max([d.get("num_threads", 1) for d in threadpool_info() if d.get("user_api") == "blas"] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    return {
        "value": steps / dt,
        "value_1thread": one_thread,
        "unit": "it/s",
        "cores": int(cores),
        "kind": "port",
        "sample": f"the full workload: oracle FFB fixed-step on the downloaded {m}x{n} {A.dtype.name} matrix "
                  f"({A.nbytes / 2**30:.1f} GiB, copied to the host in {t_dl:.1f} s), {steps} iterations in {dt:.1f} s",
    }


def cpu_baseline(m, n, sample_cols, steps, seed):
    """Reference op sequence on the host cores (oracle = numpy/OpenBLAS restatement), bounded sample."""
    import numpy as np

    from oracle import proxgrad_oracle as o

    ns = min(n, sample_cols)
    rng = np.random.default_rng(seed)
    A = np.asfortranarray(rng.standard_normal((m, ns), dtype=np.float32) / np.float32(math.sqrt(m)))
    xt = np.zeros(ns, np.float32)
    k = max(1, ns // 1000)
    xt[rng.choice(ns, k, replace=False)] = rng.standard_normal(k).astype(np.float32)
    b = A @ xt + np.float32(0.01) * rng.standard_normal(m).astype(np.float32)
    lam = np.float32(0.1) * np.float32(np.max(np.abs(A.T @ b)))
    v = np.ones(ns, np.float32) / np.float32(math.sqrt(ns))
    for _ in range(10):
        v = A.T @ (A @ v)
        v /= np.linalg.norm(v)
    Lf = np.float32(1.1 * np.linalg.norm(A @ v) ** 2)
    it = iter(o.FastForwardBackwardIteration(f=o.LeastSquares(A, b), g=o.NormL1(lam), x0=np.zeros(ns, np.float32), Lf=Lf))
    next(it)
    next(it)  # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        next(it)
    dt = time.perf_counter() - t0
    its_sample = steps / dt
    one_thread = None
    try:
        from threadpoolctl import threadpool_info, threadpool_limits

        cores = max([d.get("num_threads", 1) for d in threadpool_info() if d.get("user_api") == "blas"] or [1])
        # the single-thread figure (the reference's runbenchmarks.jl pins BLAS to one thread)
        with threadpool_limits(limits=1, user_api="blas"):
            s1 = max(2, steps // 5)
            t0 = time.perf_counter()
            for _ in range(s1):
                next(it)
            one_thread = s1 / (time.perf_counter() - t0) * ns / n
    except Exception:
        cores = os.cpu_count() or 1
    return {
        "value": its_sample * ns / n,
        "value_1thread": one_thread,
        "unit": "it/s",
        "cores": int(cores),
        "kind": "port",
        "sample": f"oracle FFB fixed-step, m={m} n={ns} f32 ({steps} it, {dt:.2f} s, {its_sample:.2f} it/s on the sample; "
                  f"scaled linearly in n to n={n})",
    }


def main():
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    import proximalalgorithms.jl_amd as pa

    # stdout carries exactly ONE line (the JSON, written last by rank 0): RCCL prints a version banner through C
    # stdio on fd 1 when a communicator is created, so everything else on fd 1 is routed to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with `python -m torch.distributed.run --nproc-per-node N`")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_comm:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        if world == 1:
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    m_glob, n = WORKLOADS[args.workload]
    m_glob = args.m or m_glob
    if args.scaling == "weak":
        m_glob *= world
    n = args.n or n
    dtype = np.float32 if args.dtype == "f32" else np.float64
    ctx = pa.get_context(local_rank)
    # N > 1: column shards keep the single-sweep iteration on every GPU (one all-reduce of m + 4 N elements per
    # iteration); row shards (north_star's layout) iterate with two sweeps and all-reduce [grad ; f] (n + 1 elements)
    sharding = args.sharding
    if sharding == "auto":
        sharding = "cols" if (world > 1 or args.force_comm) and args.sweeps == "one" and args.scaling == "strong" else "rows"
    if world == 1 and not args.force_comm:
        sharding = "none"
    cols = sharding == "cols"
    if cols:
        row_off, m_loc = 0, m_glob
        col_off, n_loc = pa.shard_cols(n, world, rank)
    else:
        row_off, m_loc = pa.shard_rows(m_glob, world, rank)
        col_off, n_loc = 0, n

    def allreduce_scalar(v, op):
        """a Python scalar reduced over the ranks (setup only)"""
        if world == 1:
            return float(v)
        t = torch.tensor([float(v)], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=op)
        return float(t.item())

    # ---------------- problem setup (untimed): A resident in HBM, b, lam, Lf ----------------
    t_setup = time.perf_counter()
    A = pa.HIPMatrix.synthetic(m_loc, n_loc, dtype, seed=args.seed, row_offset=row_off, col_offset=col_off, m_global=m_glob,
                               ctx=ctx)
    rng = np.random.default_rng(args.seed + 12345)
    k = max(1, n // 1000)
    x_true = np.zeros(n, dtype)
    x_true[rng.choice(n, size=k, replace=False)] = rng.standard_normal(k).astype(dtype)
    noise = np.random.default_rng(args.seed + 54321).standard_normal(m_glob).astype(dtype)[row_off:row_off + m_loc]
    b = A.mul(pa.HIPVector.from_numpy(x_true[col_off:col_off + n_loc], ctx))  # row shards: rows are independent
    if cols and world > 1:
        pa.allreduce_sum_(b.torch())  # column shards: b = sum_p A[:, J_p] x_true[J_p]
    b.axpby_(1.0, b, 0.01, pa.HIPVector.from_numpy(noise, ctx))
    comm = None
    if world > 1 or args.force_comm:
        comm = (pa.NativeRcclComm(overlap=args.overlap, shard="cols" if cols else "rows") if args.collective == "native"
                else pa.TorchDistributedComm(overlap=args.overlap, shard="cols" if cols else "rows"))
    f = pa.LeastSquares(A, b, comm=comm)
    zero_n = pa.HIPVector.zeros(n_loc, dtype, ctx)
    _, g0 = f.value_and_gradient(zero_n)  # = -A'b (row shards: all-reduced; column shards: this rank's columns)
    g0_inf = float(g0.norm_inf())
    if cols:
        g0_inf = allreduce_scalar(g0_inf, dist.ReduceOp.MAX)
    lam = dtype(0.1) * dtype(g0_inf)  # test_lasso_small.jl:29
    Lf = None
    if args.mode == "fixed":
        f0 = pa.LeastSquares(A, pa.HIPVector.zeros(m_loc, dtype, ctx), comm=comm)  # x -> A'A x
        v = pa.HIPVector.zeros(n_loc, dtype, ctx).fill_(1.0 / math.sqrt(n))
        w = v.similar()
        nrm = dtype(1)
        for _ in range(30):
            f0.value_and_gradient(v, out=w)
            nrm2 = float(w.norm()) ** 2
            if cols:
                nrm2 = allreduce_scalar(nrm2, dist.ReduceOp.SUM)
            nrm = dtype(math.sqrt(nrm2))
            v.axpby_(1.0 / float(nrm), w)
        Lf = dtype(1.1) * nrm  # ||A||^2 estimate (+10 % margin: power iteration under-estimates)
        del f0
    ctx.sync()
    t_setup = time.perf_counter() - t_setup

    iteration = pa.FastForwardBackwardIteration(f=f, g=pa.NormL1(lam), x0=zero_n, Lf=Lf, single_sweep=args.sweeps == "one")
    it = iter(iteration)
    state = next(it)  # init (k = 1)
    stop_rule = lambda s: float(s.res_inf) / float(s.gamma) <= 1e-6  # benchmarks.jl:57 (evaluated, not acted on)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        state = next(it)
        stop_rule(state)
    passes0 = iteration.counters.get("a_passes", 0)
    # HIP event pairs around the two GEMV kernels only: each pair is a marker packet on the stream, and the roofline
    # leg needs nothing else
    ctx.profile(args.kernel_events != "none",
                kernels=None if args.kernel_events == "all" else ("gemv_n_partial", "gemv_t", "gemv_tn"))
    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        state = next(it)
        stop_rule(state)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = ctx.profile_read()
    ctx.profile(False)
    a_passes = iteration.counters.get("a_passes", 0) - passes0

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    its = args.steps / elapsed
    es = 4 if args.dtype == "f32" else 8
    sweeps = a_passes / max(args.steps, 1)  # reads of A per iteration actually executed
    # the algorithmic figure counts the passes the mode REQUIRES when A x and A' r are separate sweeps (2 for
    # fixed-step FB / FFB and for adaptive FFB with the residual pair); the single-sweep iteration moves fewer bytes --
    # both are reported, labelled
    passes_alg = max(2.0, sweeps) if args.sweeps == "two" or (world > 1 and not cols) else 2.0
    bytes_iter_local = passes_alg * m_loc * n_loc * es + 10 * n_loc * es + 3 * m_loc * es
    bytes_moved_local = sweeps * m_loc * n_loc * es + 10 * n_loc * es + 3 * m_loc * es
    # dominant kernel = the slowest sweep over A; algorithmic bytes of one launch = the local A block + its vectors
    kern = {}
    n_cnt = prof["gemv_n_partial"][0]
    for name, vec_bytes in (("gemv_n_partial", n_loc * es), ("gemv_t", m_loc * es + n_loc * es),
                            ("gemv_tn", (m_loc + 7 * n_loc) * es)):
        cnt, ms = prof[name]
        if cnt:
            avg_ms = ms / cnt
            # with a collective attached pass T runs as several column-chunk launches per evaluation
            # (pg_gemv.hip ls_grad_stage_t): one launch then covers 1/chunks of the local block
            evals = max(a_passes - n_cnt - prof["gemv_tn"][0], 1) if name == "gemv_t" else cnt
            launch_bytes = (m_loc * n_loc * es + vec_bytes) * evals / cnt
            kern[name] = {"launches": cnt, "avg_ms": avg_ms, "bytes": launch_bytes,
                          "launches_per_pass": cnt / evals, "GBps": launch_bytes / (avg_ms * 1e-3) / 1e9}
    dom = max(kern, key=lambda k_: kern[k_]["avg_ms"]) if kern else None
    roofline = None
    if dom:
        # HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc cannot run inside this process);
        # only quoted when they were collected on this exact workload
        traffic, traffic_src = None, None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            rec = pmc.get(args.workload)
            if rec and world == 1 and args.m is None and args.n is None and args.dtype == "f32":
                traffic = rec["kernels"][dom]["hbm_bytes"]
                traffic_src = rec["source"]
        except Exception:
            pass
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(kern[dom]["GBps"], 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(kern[dom]["GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_source": traffic_src,
                    "avg_launch_ms": round(kern[dom]["avg_ms"], 4), "launches": kern[dom]["launches"],
                    "algorithmic_bytes_per_launch": int(kern[dom]["bytes"]),
                    "per_kernel": {k_: {"avg_ms": round(v["avg_ms"], 4), "GBps": round(v["GBps"], 1),
                                        "launches": v["launches"], "launches_per_pass": round(v["launches_per_pass"], 2)}
                                   for k_, v in kern.items()},
                    "whole_iteration": {"algorithmic_bytes_per_gpu": int(bytes_iter_local),
                                        "GBps_per_gpu": round(bytes_iter_local * its / 1e9, 1),
                                        "frac": round(bytes_iter_local * its / 1e9 / HBM_PEAK_GBS, 4),
                                        "sweeps_of_A_per_iteration": round(sweeps, 3),
                                        "hbm_bytes_moved_per_gpu": int(bytes_moved_local),
                                        "hbm_GBps_moved_per_gpu": round(bytes_moved_local * its / 1e9, 1),
                                        "frac_of_bytes_moved": round(bytes_moved_local * its / 1e9 / HBM_PEAK_GBS, 4),
                                        "note": "algorithmic_bytes = SURVEY 8(d): A x and A' r as separate passes (2 m n s "
                                                "+ vectors); frac > 1 means the iteration moves fewer bytes than that (the "
                                                "single sweep reads A once); *_moved = bytes actually read/written"}}

    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            # the whole matrix when the host can hold it (3x headroom), else a column sample scaled linearly in n
            need = 3 * m_glob * n * es
            lim = _host_memory_limit()
            if args.cpu_baseline == "full" or (args.cpu_baseline == "auto" and lim is not None and lim >= need):
                cpu = cpu_baseline_full(A, b, lam, Lf)
                # single-thread figure from the column sample (a one-thread pass over the whole matrix takes too long)
                cpu["value_1thread"] = cpu_baseline(m_glob, n, args.cpu_sample_cols, max(2, args.cpu_steps // 2),
                                                    args.seed)["value_1thread"]
            else:
                cpu = cpu_baseline(m_glob, n, args.cpu_sample_cols, args.cpu_steps, args.seed)
        line = {
            "metric": "FastForwardBackward iters/sec on LASSO (m=%d, n=%d, %s)" % (m_glob, n, args.dtype),
            "value": round(its, 4),
            "unit": "it/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "FFB LASSO m=%d n=%d %s, %s step, %s of A sharded over %d GPU(s)"
                                   % (m_glob, n, "Float32" if args.dtype == "f32" else "Float64", args.mode,
                                      "columns" if cols else "rows", world),
                       "m": m_glob, "n": n, "mode": args.mode, "sharding": sharding, "shards": world,
                       "row_shards": 1 if cols else world, "m_per_gpu": m_loc, "n_per_gpu": n_loc,
                       "lambda": float(lam), "Lf": float(Lf) if Lf is not None else None, "seed": args.seed,
                       "a_passes_per_step": a_passes / max(args.steps, 1), "sweeps": args.sweeps if (world == 1 or cols) else "two",
                       "setup_s": round(t_setup, 2),
                       "final": {"gamma": float(state.gamma), "f_x": float(state.f_x), "g_z": float(state.g_z),
                                 "res_inf_over_gamma": float(state.res_inf) / float(state.gamma)}},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
    else:
        line = None
    if world > 1 or args.force_comm:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if line is not None:
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
