#!/usr/bin/env python3
"""bench.py -- FastForwardBackward iterations/sec on synthetic LASSO (BASELINE.json metric) on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is ONE FastForwardBackward iteration (fast_forward_backward.jl:106-145) of the fused HIP engine on the
headline workload  m = 16384, n = 2^20, Float32, fixed step gamma = 1/Lf  (BASELINE.json north_star).  A (64 GiB)
is generated on the device and is resident in HBM before the timed region.  An iteration reads A ONCE (the
single-sweep iteration: A' r, prox, next extrapolation and next residual per column while it is in registers);
--sweeps two runs A x and A' r as separate sweeps like the reference.

N > 1: one process per GPU over RCCL (torch.distributed backend "nccl").  Started as plain `python bench.py --gpus N`
the script launches `python -m torch.distributed.run --nproc-per-node N` itself as a CHILD process (before anything
touches the GPU), relays the child's JSON line and exits with its code; under torch.distributed.run it is a rank.
Rank 0 prints ONE JSON line.  Its top-level value is the STRONG-scaled headline problem in north_star's layout: ROW blocks of
A per rank.  It is measured first in the form that cannot fail on a new fabric -- two sweeps per iteration and one RCCL
all-reduce of [grad (n) ; f] per gradient evaluation (benchmark/benchmarks.jl:15-16 per shard) -- and then UPGRADED to the
row-team form (one read of A per iteration: the ranks exchange per-column partial dots through each other's inbox inside the
sweep) when that form, run for the same K steps in a process group of its own, passes its attach-time self-test on every
rank and loses no sweep to a bounded wait; `config.row_layout` says which form the value is ("row_teams" | "two_sweeps") and
`config.row_layout_reason` why.  The line also carries
  rows_two_sweeps    (after an upgrade) the two-sweep record the line would have had,
  cols_strong        the same problem with COLUMN blocks (not the contract layout: every GPU keeps the single sweep, one
                     all-reduce of m + 8 (N + 1) floats per iteration),
  config5_weak_rows  BASELINE config 5 and its smaller twins: 16384 rows PER GPU (131072 x 2^20 at N = 8), row blocks,
  config5_weak_cols  the same global problem with column blocks (each GPU: m x n/N, the long-column single sweep),
  config5_weak_rows_teams  config 5 as a row team,
each with its own roofline, the world size RCCL reports and the all-reduce payload.

N = 1: the line also carries `also`: the reference benchmark's own adaptive mode on the headline matrix
(benchmark/benchmarks.jl:55-61), BASELINE configs 2, 3 and 4 and two per-GPU block shapes at N = 8 (long and short columns), each with
its roofline (--no-also skips them).

Extra legs in the same line:
  sustained     the main record's iteration kept running for --sustain seconds (one-GPU headline: 5) after the K timed steps:
                it/s over that window (a burst check, and a window long enough for an outside GPU-busy sampler)
  roofline      HBM roofline of the dominant kernel (the slowest sweep over A), timed live with HIP event pairs
                on the launch stream (pg_ctx_profile_*), algorithmic bytes = one full read of the local A block +
                its vectors; whole_iteration reports the SURVEY 8(d) two-pass figure and the bytes actually moved.
                traffic = HBM bytes per launch from the committed rocprofv3 --pmc passes, quoted only while the
                kernel sources still hash to what the passes were taken on (else null + traffic_stale).
  in_library_loop  (fixed step, one rank) the same K iterations enqueued by the library without a host round trip in between
                (pg_iter_run_batched); `value` stays the stepped loop of the reference's iterator
  cpu_baseline  the CPU restatement (oracle/csrc/cpu_twin.c: C / OpenMP, same unfused op order as the reference; the
                numpy + OpenBLAS oracle's rate beside it) timed on this host on the SAME workload (the device matrix copied
                to host memory) when memory allows, else on a bounded column sample scaled to it/s of the full workload.
  also_settle_s seconds the extra records waited (untimed) for the driver to finish clearing device memory freed just
                before them: kernels running during that clearing lose 2-4 % (profiles/r3_freed_memory_settle.md)
"""
import argparse
import ctypes
import hashlib
import json
import math
import os
import signal
import socket
import subprocess
import sys
import threading
import time
import traceback

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
# the sources the sweep kernels are built from: the PMC passes in profiles/pmc_traffic.json are tied to their hash
KERNEL_SOURCES = ("pg_gemv_tn.h", "pg_gemv_tnt.h", "pg_cgmap.h", "pg_lanes.h", "pg_gemv.hip", "pg_gemv_tn2.hip", "pg_gemv_tn3.hip")


def parse_args(argv=None):
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
                   help="one: the single-sweep iteration (A read once per iteration) -- two: A x and A' r as "
                        "separate sweeps like the reference (always the case when rows are sharded)")
    p.add_argument("--sharding", choices=["auto", "rows", "cols"], default="auto",
                   help="N > 1: layout of the TOP-LEVEL record (auto = rows: north_star's contract, upgraded to row teams when they "
                        "run clean -- see the module docstring); the other layout is reported as a sub-record")
    p.add_argument("--row-teams", action="store_true",
                   help="with --sharding rows: the top-level record itself runs as a row team, in this process group (the ranks "
                        "exchange per-column partial dots through each other's inbox inside the sweep: one read of A per iteration)")
    p.add_argument("--no-row-teams", action="store_true", help="N > 1: no row-team records (the top-level row record stays on two sweeps)")
    p.add_argument("--row-teams-child", action="store_true", help=argparse.SUPPRESS)  # the isolated process of those two records
    p.add_argument("--no-also", action="store_true",
                   help="skip the extra records (N = 1: adaptive headline + configs 2 / 3 / 4; N > 1: the other layouts)")
    p.add_argument("--sustain", type=float, default=None,
                   help="seconds the main record's iteration keeps running after the K timed steps (reported as `sustained`); "
                        "default: 5 for the one-GPU headline line, 0 otherwise")
    p.add_argument("--also-budget", type=float, default=150.0, help="seconds the extra records may take in total (N = 1)")
    p.add_argument("--no-settle", dest="settle", action="store_false",
                   help="do not wait for the driver's background clearing of freed device memory before an extra record's set-up")
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
                   help="TOP-LEVEL record: strong = the global problem is fixed and split over the ranks (default); weak = "
                        "every rank holds --m rows (BASELINE config 5 = --scaling weak --m 16384 on 8 GPUs)")
    p.add_argument("--collective", choices=["auto", "torch", "native"], default="auto",
                   help="N > 1: all-reduce through the library's own RCCL communicator (native: no Python in the collective path; the "
                        "default when librccl loads and the backend is nccl) or through torch.distributed (torch)")
    p.add_argument("--overlap", action="store_true",
                   help="pipeline the [grad ; f] all-reduce with pass T in column chunks (N > 1, row blocks). Off by default: at "
                        "the headline shard shape the chunking costs ~60 us/step, about what it can hide (DESIGN.md section 6)")
    p.add_argument("--force-comm", action="store_true",
                   help="diagnostic: attach the collective even with one rank (measures the cost of the sharded code path)")
    p.add_argument("--share-device", action="store_true",
                   help="functional test mode: every rank uses cuda:0 (e.g. 2 ranks on a 1-GPU box, with --backend gloo)")
    p.add_argument("--launch-timeout", type=float, default=900.0,
                   help="`python bench.py --gpus N` without a launcher: wall-clock limit of the child job; on expiry its process "
                        "group is ended and an error JSON line is printed (exit code 124)")
    p.add_argument("--record-timeout", type=float, default=300.0,
                   help="deadline of the top-level record (setup + warm-up + timed steps), seconds; on expiry rank 0 prints the "
                        "line it has (with `error` and `stage`) and every rank exits")
    p.add_argument("--sub-record-timeout", type=float, default=240.0, help="deadline of each further record, seconds")
    p.add_argument("--stall-timeout", type=float, default=90.0,
                   help="a record that makes no progress (no iteration, no setup phase finished) for this long is treated like an "
                        "expired deadline; kept below --init-timeout so that this script, not the collective's own watchdog, ends the job")
    p.add_argument("--init-timeout", type=float, default=120.0, help="torch.distributed.init_process_group timeout, seconds")
    p.add_argument("--inject-fault", default=None, metavar="RANK:STAGE:KIND",
                   help="test hook: when RANK enters the record STAGE it hangs (KIND = hang) or exits with code 1 (KIND = exit)")
    if argv is None and len(sys.argv) == 1 and os.environ.get("PG_BENCH_ARGV") and os.environ.get("WORLD_SIZE"):
        argv = json.loads(os.environ["PG_BENCH_ARGV"])  # a rank started by self_launch: see launch_command
    return p.parse_args(argv)


def metric_name(args, world=None):
    m_base, n = WORKLOADS[args.workload]
    m_base, n = args.m or m_base, args.n or n
    world = world or args.gpus
    return "FastForwardBackward iters/sec on LASSO (m=%d, n=%d, %s)" % (m_base * world if args.scaling == "weak" else m_base, n, args.dtype)


def error_line(args, error, stage, **more):
    """the ONE JSON line of a run that could not measure its top-level record"""
    d = {"metric": metric_name(args), "value": None, "unit": "it/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
         "ms_per_step": None, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": args.dtype,
         "data": "synthetic", "error": error, "stage": stage}
    d.update(more)
    return d


# ---------------------------------------------------------------------------------------------------------------
# N > 1 from a plain shell: start the ranks as a child process (never exec: see the module docstring)
# ---------------------------------------------------------------------------------------------------------------
def launch_command(args, port):
    # the script's own arguments travel in PG_BENCH_ARGV (self_launch), not on the launcher's command line: torch.distributed.run
    # parses with abbreviations and takes e.g. `--m 4096` for an ambiguous option of its own
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]


def _find_line(text):
    line = None
    for ln in text.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                if "metric" in json.loads(ln):
                    line = ln
            except ValueError:
                pass
    return line


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    as a CHILD in its own process group (never exec: nothing in this process has touched the GPU), relay its JSON line and
    return its exit code.  The child gets --launch-timeout seconds of wall clock: on expiry the group gets SIGTERM (rank 0
    then prints the line it has), 10 s later SIGKILL.  Whatever happens, stdout carries ONE JSON line: the ranks' own, or
    an error line {"metric", "value": null, "error", "stage", "stderr_tail"} built here."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = launch_command(args, port)
    env = dict(os.environ)
    env["PG_BENCH_ARGV"] = json.dumps(sys.argv[1:])
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(args.gpus, 1))))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, start_new_session=True)
    out_chunks, err_tail = [], []

    def pump_out():
        for chunk in iter(lambda: proc.stdout.read(65536), b""):
            out_chunks.append(chunk)

    def pump_err():  # relayed as it comes; the last lines are kept for the error line
        for raw in iter(proc.stderr.readline, b""):
            sys.stderr.buffer.write(raw)
            sys.stderr.flush()
            err_tail.append(raw.decode(errors="replace").rstrip())
            del err_tail[:-60]

    threads = [threading.Thread(target=pump_out, daemon=True), threading.Thread(target=pump_err, daemon=True)]
    for t in threads:
        t.start()
    timed_out = False
    try:
        rc = proc.wait(timeout=args.launch_timeout)
    except subprocess.TimeoutExpired:
        timed_out = True
        for sig, grace in ((signal.SIGTERM, 10.0), (signal.SIGKILL, 5.0)):
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                continue
        rc = 124
    for t in threads:
        t.join(timeout=5.0)
    line = _find_line(b"".join(out_chunks).decode(errors="replace"))
    tail = [ln for ln in err_tail if ln.strip()][-12:]
    if line is None:
        if timed_out:
            d = error_line(args, "timeout", "launch: no line from the ranks within %.0f s" % args.launch_timeout, stderr_tail=tail)
        elif rc != 0:
            d = error_line(args, "the ranks exited with code %d without a line" % rc, "launch", stderr_tail=tail)
        else:
            d = error_line(args, "the ranks exited without a JSON line", "launch", stderr_tail=tail)
            rc = 1
        line = json.dumps(d)
    elif timed_out:
        d = json.loads(line)
        d.setdefault("error", "timeout")
        d.setdefault("stage", "launch: ended after %.0f s" % args.launch_timeout)
        line = json.dumps(d)
    sys.stdout.write(line + "\n")
    sys.stdout.flush()
    return rc


# ---------------------------------------------------------------------------------------------------------------
# deadlines inside a rank: a hang in one record must not cost the records already measured
# ---------------------------------------------------------------------------------------------------------------
class Watchdog:
    """Every record runs under a deadline and a stall detector.  A daemon thread watches both; when one expires rank 0
    writes the line it has -- the records measured so far, `error` and `stage` -- and every rank leaves with os._exit
    (0 when the top-level record was measured, else 3; ranks > 0 wait `grace` seconds so that rank 0's line is out before
    the launcher reacts to the first exit).  SIGTERM (the launcher ending the job because another rank died) takes the
    same path through a sigwait thread, which works while the main thread is stuck inside a collective."""

    def __init__(self, rank, stall_s, write_line, grace=3.0, inject=None):
        self.rank, self.stall_s, self.write_line, self.grace = rank, stall_s, write_line, grace
        self.lock = threading.Lock()
        self.stage, self.deadline_s = "startup", float("inf")
        self.t_stage = self.t_beat = time.monotonic()
        self.main_done = False
        self.closed = False
        self.inject = None
        if inject:
            r, st, kind = inject.split(":")
            if int(r) == rank:
                self.inject = (st, kind)
        threading.Thread(target=self._watch, daemon=True).start()
        threading.Thread(target=self._sigwait, daemon=True).start()

    def enter(self, stage, deadline_s, stall=True):
        self.stage, self.deadline_s = stage, deadline_s
        self.stall_on = stall
        self.t_stage = self.t_beat = time.monotonic()
        if self.inject and self.inject[0] == stage:
            if self.inject[1] == "exit":
                sys.stderr.write("bench.py: injected fault: rank %d exits in stage %s\n" % (self.rank, stage))
                sys.stderr.flush()
                os._exit(1)
            while True:  # "hang": the main thread never comes back (the watchdog thread ends the process)
                time.sleep(1.0)

    def beat(self):
        self.t_beat = time.monotonic()

    def close(self):
        """the normal end: from here on the main thread owns the line"""
        with self.lock:
            if self.closed:  # a watchdog exit is under way
                time.sleep(3600)
            self.closed = True

    def _fire(self, error):
        with self.lock:
            if self.closed:
                return
            self.closed = True
        code = 0 if self.main_done else 3
        sys.stderr.write("bench.py: rank %d: %s in stage %s -> exit %d\n" % (self.rank, error, self.stage, code))
        sys.stderr.flush()
        if self.rank == 0:
            try:
                self.write_line(error, self.stage)
            except Exception:
                traceback.print_exc()
        else:
            time.sleep(self.grace)
        os._exit(code)

    def _watch(self):
        while True:
            time.sleep(0.25)
            now = time.monotonic()
            if now - self.t_stage > self.deadline_s:
                self._fire("timeout: record deadline of %.0f s expired" % self.deadline_s)
            if getattr(self, "stall_on", True) and now - self.t_beat > self.stall_s:
                self._fire("timeout: no progress for %.0f s" % self.stall_s)

    def _sigwait(self):
        signal.sigwait({signal.SIGTERM})
        self._fire("terminated by the launcher (SIGTERM: another rank failed or the job ran out of time)")


# ---------------------------------------------------------------------------------------------------------------
# CPU leg
# ---------------------------------------------------------------------------------------------------------------
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


def _effective_cpus():
    """CPUs this process can actually keep busy: the affinity mask capped by the cgroup's CPU quota (cpu.max = quota period).
    The GPU box shows 256 hardware threads but grants 16 CPUs' worth of time; more runnable threads than that are throttled
    and a streaming loop gets SLOWER (measured there: 16 threads 140-160 GB/s, 64 threads 70, 128 threads 43)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(math.ceil(float(quota) / float(period)))))
    except Exception:
        pass
    return max(1, n)


def _blas_threads():
    try:
        from threadpoolctl import threadpool_info

        return max([d.get("num_threads", 1) for d in threadpool_info() if d.get("user_api") == "blas"] or [1])
    except Exception:
        return os.cpu_count() or 1


def host_stream_gbps(threads=None, mib_per_array=1024, reps=3):
    """The host's own streaming rate, so that the CPU leg's it/s can be read as a fraction of what this host can move:
    STREAM "add" (a = b + c, Float32, 12 bytes per element by STREAM's count) with numpy kernels on `threads` Python threads
    (numpy releases the GIL inside the loop), each on its own contiguous chunk, sustained over at least one second.  `threads`
    defaults to the CPUs the cgroup grants (_effective_cpus).  Returns (GB/s with all threads, GB/s with one thread, threads)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor

    threads = threads or min(_effective_cpus(), 64)
    n = mib_per_array * (1 << 20) // 4
    b, c_, a = np.ones(n, np.float32), np.ones(n, np.float32), np.empty(n, np.float32)
    a[:] = 0  # touch every page before timing

    def rate(T):
        # sustained, not a burst: passes back to back for at least a second (a CPU quota refills per 100 ms period, so a
        # single 15 ms pass on many threads would show a rate the cgroup does not sustain)
        cuts = [n * k // T for k in range(T + 1)]
        with ThreadPoolExecutor(T) as ex:
            one = lambda: list(ex.map(lambda k: np.add(b[cuts[k]:cuts[k + 1]], c_[cuts[k]:cuts[k + 1]], out=a[cuts[k]:cuts[k + 1]]), range(T)))
            one()
            passes, t0 = 0, time.perf_counter()
            while passes < reps or time.perf_counter() - t0 < 1.0:
                one()
                passes += 1
            dt = time.perf_counter() - t0
        return 3 * n * 4 * passes / dt / 1e9

    return rate(threads), rate(1), threads


def _cpu_bandwidth_fields(rec, m, n, es):
    """what the it/s of the CPU leg means in bytes: the oracle streams A twice per iteration (A x, then A' r)"""
    rec["achieved_GBps"] = round(2.0 * m * n * es * rec["value"] / 1e9, 1)
    if rec.get("host_read_GBps") is not None:
        # a ceiling only if it bounds what it is compared with: say which it is on this host
        # (within 2 %: two measurements of the same memory system a few seconds apart)
        rec["host_read_is_ceiling"] = bool(rec["host_read_GBps"] >= 0.98 * rec["achieved_GBps"])
        rec["host_read_note"] += ("; the host's read ceiling for this job: achieved_GBps / host_read_GBps is the CPU leg's own roofline fraction"
                                  if rec["host_read_is_ceiling"] else
                                  "; NOT a ceiling on this host (the iteration's own passes stream faster): read it as a second rate, no more")
    if rec.get("value_1thread"):
        rec["achieved_GBps_1thread"] = round(2.0 * m * n * es * rec["value_1thread"] / 1e9, 1)
    try:
        g_all, g_one, T = host_stream_gbps()
        rec["host_stream_GBps"] = round(g_all, 1)
        rec["host_stream_GBps_1thread"] = round(g_one, 1)
        rec["host_cpus_granted"] = _effective_cpus()
        rec["host_stream_note"] = "STREAM add (Float32, numpy kernels on %d threads = the CPUs the cgroup grants, 3 x 1 GiB arrays, sustained " \
                                  "over >= 1 s; numpy's add pays a write-allocate, so a read-only pass can run faster: host_read_GBps)" % T
    except Exception as e:  # never let the side measurement cost the line
        rec["host_stream_GBps"] = None
        rec["host_stream_note"] = "not measured: %s" % str(e)[:120]
    return rec


def cpu_baseline_full(A_dev, b_dev, lam, Lf, budget_s=25.0, max_steps=8):
    """The SAME workload on the host cores: the device matrix is copied to host memory (a few seconds over PCIe) and the CPU
    restatements of the reference's op sequence step on it -- measured on the full matrix, not extrapolated:
      value           the C / OpenMP twin (oracle/csrc/cpu_twin.c: unfused A x and A' r, each threaded over all cores), Float32;
      value_1thread   the same twin on one thread (the reference's runbenchmarks.jl pins BLAS to one thread), one iteration;
      numpy_openblas  the numpy oracle (what the parity tests compare with) on the same matrix: OpenBLAS' sgemv scales
                      poorly past a few threads, which is why it is not the headline CPU figure."""
    import numpy as np

    from oracle import proxgrad_oracle as o

    t0 = time.perf_counter()
    A = None
    try:  # pages of the host copy placed by the threads that will stream them (a multi-socket host)
        from oracle import cpu_twin

        m_, n_ = A_dev.shape
        if A_dev.dtype == np.float32:
            A = cpu_twin.first_touch(np.empty((m_, n_), np.float32, order="F"), threads=_effective_cpus())
    except Exception:
        A = None
    A = A_dev.numpy(out=A)
    b = b_dev.numpy()
    t_dl = time.perf_counter() - t0
    m, n = A.shape
    es = A.dtype.itemsize
    rec = None
    if A.dtype == np.float32:
        try:
            from oracle import cpu_twin

            ncpu = _effective_cpus()  # (not omp_get_max_threads: the cgroup's CPU quota, see _effective_cpus)
            _, _, sec1, thr = cpu_twin.ffb(A, b, lam, Lf, 1, threads=ncpu)  # one iteration to size the run
            steps = int(max(2, min(max_steps, 0.5 * budget_s / max(sec1, 1e-3))))
            _, _, sec, thr = cpu_twin.ffb(A, b, lam, Lf, steps, threads=ncpu)
            _, _, sec_1t, _ = cpu_twin.ffb(A, b, lam, Lf, 1, threads=1)
            read_all = cpu_twin.read_gbps(A, threads=ncpu)  # this host's read rate on the same 64 GiB, same threads
            cpu_twin.load().cpu_twin_set_threads(ncpu)
            rec = {"host_read_GBps": round(read_all, 1),
                   "host_read_note": "OpenMP passes summing the same matrix on the same threads, pages first-touched by the threads that read "
                                     "them; the better of two access patterns (eight interleaved streams per thread / one contiguous stream "
                                     "per thread: oracle/csrc/cpu_twin.c::cpu_twin_read_pass, _seq)",
                   "value": steps / sec, "value_1thread": 1.0 / sec_1t, "unit": "it/s", "cores": int(thr), "kind": "port",
                   "impl": "C / OpenMP twin of the reference's unfused op sequence (oracle/csrc/cpu_twin.c)",
                   "sample": f"the full workload: FFB fixed-step on the downloaded {m}x{n} float32 matrix ({A.nbytes / 2**30:.1f} GiB, "
                             f"copied to the host in {t_dl:.1f} s), {steps} iterations in {sec:.1f} s on {thr} threads; 1 thread: "
                             f"1 iteration in {sec_1t:.1f} s"}
        except Exception as e:  # no compiler on this host, ...: the numpy figures below become the record
            sys.stderr.write("bench.py: C/OpenMP CPU twin unavailable (%s); the CPU leg uses the numpy oracle\n" % str(e)[:200])
    it = iter(o.FastForwardBackwardIteration(f=o.LeastSquares(A, b), g=o.NormL1(lam), x0=np.zeros(n, A.dtype), Lf=Lf))
    next(it)  # init (two passes), untimed like the GPU side
    np_budget = budget_s if rec is None else 0.35 * budget_s
    import contextlib

    try:  # the BLAS pool no larger than the CPUs the cgroup grants (an oversubscribed pool is throttled, not faster)
        from threadpoolctl import threadpool_limits

        blas_cap = threadpool_limits(limits=min(int(_blas_threads()), _effective_cpus()), user_api="blas")
    except Exception:
        blas_cap = contextlib.nullcontext()
    with blas_cap:
        blas_n = int(_blas_threads())
        steps_np, t0 = 0, time.perf_counter()
        while steps_np < max_steps and (steps_np < 2 or time.perf_counter() - t0 < np_budget):
            next(it)
            steps_np += 1
        dt = time.perf_counter() - t0
    numpy_rec = {"value": steps_np / dt, "cores": blas_n, "achieved_GBps": round(2.0 * m * n * es * steps_np / dt / 1e9, 1),
                 "sample": f"{steps_np} iterations in {dt:.1f} s on {blas_n} BLAS threads"}
    if rec is None:
        one_thread, note1 = None, ""
        try:
            from threadpoolctl import threadpool_limits

            with threadpool_limits(limits=1, user_api="blas"):
                t1 = time.perf_counter()
                next(it)
                d1 = time.perf_counter() - t1
            one_thread = 1.0 / d1
            note1 = f"; 1 BLAS thread: 1 iteration on the same matrix in {d1:.1f} s"
        except Exception:
            pass
        rec = {"value": steps_np / dt, "value_1thread": one_thread, "unit": "it/s", "cores": blas_n, "kind": "port",
               "impl": "numpy / OpenBLAS oracle (oracle/proxgrad_oracle.py)",
               "sample": f"the full workload: oracle FFB fixed-step on the downloaded {m}x{n} {A.dtype.name} matrix "
                         f"({A.nbytes / 2**30:.1f} GiB, copied to the host in {t_dl:.1f} s), {steps_np} iterations in {dt:.1f} s{note1}"}
    rec["numpy_openblas"] = numpy_rec
    if rec.get("impl", "").startswith("C / OpenMP") and numpy_rec["value"] > rec["value"]:
        # `value` is the FASTER of the two CPU implementations on this host (which one wins differs from box to box:
        # profiles/r3_bench_default.json and its predecessors); the other stays in the record
        rec["c_openmp_twin"] = {"value": rec["value"], "cores": rec["cores"], "sample": rec["sample"]}
        rec["value"], rec["cores"] = numpy_rec["value"], blas_n
        rec["impl"] = "numpy / OpenBLAS oracle (oracle/proxgrad_oracle.py): faster on this host than the C / OpenMP twin (c_openmp_twin)"
        rec["sample"] = (f"the full workload: FFB fixed-step on the downloaded {m}x{n} float32 matrix ({A.nbytes / 2**30:.1f} GiB, copied to the "
                         f"host in {t_dl:.1f} s), " + numpy_rec["sample"] + "; value_1thread: the C / OpenMP twin on one thread")
    del it, A, b
    return _cpu_bandwidth_fields(rec, m, n, es)


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
        from threadpoolctl import threadpool_limits

        with threadpool_limits(limits=1, user_api="blas"):
            s1 = max(2, steps // 5)
            t0 = time.perf_counter()
            for _ in range(s1):
                next(it)
            one_thread = s1 / (time.perf_counter() - t0) * ns / n
    except Exception:
        pass
    rec = {
        "value": its_sample * ns / n,
        "value_1thread": one_thread,
        "unit": "it/s",
        "cores": int(_blas_threads()),
        "kind": "port",
        "sample": f"oracle FFB fixed-step, m={m} n={ns} f32 ({steps} it, {dt:.2f} s, {its_sample:.2f} it/s on the sample; "
                  f"scaled linearly in n to n={n}; the host cannot hold the full matrix)",
    }
    del it, A
    return _cpu_bandwidth_fields(rec, m, n, 4)


# ---------------------------------------------------------------------------------------------------------------
# PMC traffic tied to the kernel sources
# ---------------------------------------------------------------------------------------------------------------
def kernel_source_hash():
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "proximalalgorithms.jl_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def pmc_traffic(workload, kernel):
    """(traffic bytes per launch | None, source | None, stale flag) from profiles/pmc_traffic.json"""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        rec = pmc.get(workload)
        if not rec or kernel not in rec.get("kernels", {}):
            return None, None, False
        if rec.get("kernel_source_sha256") != kernel_source_hash():
            return None, rec.get("source"), True
        return rec["kernels"][kernel]["hbm_bytes"], rec.get("source"), False
    except Exception:
        return None, None, False


# ---------------------------------------------------------------------------------------------------------------
# one FastForwardBackward record
# ---------------------------------------------------------------------------------------------------------------
class Dist:
    """what the records need to know about the job"""

    def __init__(self, world, rank, local_rank, backend, collective, overlap, force_comm):
        self.world, self.rank, self.local_rank = world, rank, local_rank
        self.backend, self.collective, self.overlap, self.force_comm = backend, collective, overlap, force_comm
        self.sharded = world > 1 or force_comm
        self.beat = lambda: None  # progress mark for the watchdog's stall detector

    def barrier(self):
        import torch
        import torch.distributed as dist

        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, v):
        if self.world == 1:
            return float(v)
        import torch
        import torch.distributed as dist

        t = torch.tensor([float(v)], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def reduce_scalar(self, v, op):
        if self.world == 1:
            return float(v)
        import torch
        import torch.distributed as dist

        t = torch.tensor([float(v)], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=op)
        return float(t.item())

    def ranks_seen(self):
        if not self.sharded:
            return 1
        import torch.distributed as dist

        return dist.get_world_size()


def setup_lasso(pa, ctx, D, m_glob, n, dtype, seed, layout, mode, row_teams=False):
    """A resident in HBM (this rank's block), b, lam = 0.1 ||A'b||_inf (test_lasso_small.jl:29), Lf (fixed step) -- untimed."""
    import numpy as np
    import torch.distributed as dist

    t0 = time.perf_counter()
    cols = layout == "cols"
    if cols:
        row_off, m_loc = 0, m_glob
        col_off, n_loc = pa.shard_cols(n, D.world, D.rank)
    elif layout == "rows":
        row_off, m_loc = pa.shard_rows(m_glob, D.world, D.rank)
        col_off, n_loc = 0, n
    else:
        row_off, m_loc, col_off, n_loc = 0, m_glob, 0, n
    A = pa.HIPMatrix.synthetic(m_loc, n_loc, dtype, seed=seed, row_offset=row_off, col_offset=col_off, m_global=m_glob,
                               ctx=ctx)
    ctx.sync()
    D.beat()
    rng = np.random.default_rng(seed + 12345)
    k = max(1, n // 1000)
    x_true = np.zeros(n, dtype)
    x_true[rng.choice(n, size=k, replace=False)] = rng.standard_normal(k).astype(dtype)
    noise = np.random.default_rng(seed + 54321).standard_normal(m_glob).astype(dtype)[row_off:row_off + m_loc]
    b = A.mul(pa.HIPVector.from_numpy(x_true[col_off:col_off + n_loc], ctx))  # row blocks: rows are independent
    if cols and D.world > 1:
        pa.allreduce_sum_(b.torch())  # column blocks: b = sum_p A[:, J_p] x_true[J_p]
    b.axpby_(1.0, b, 0.01, pa.HIPVector.from_numpy(noise, ctx))
    comm = None
    if layout != "none":
        shard = "cols" if cols else "rows"
        comm = (pa.NativeRcclComm(overlap=D.overlap, shard=shard) if D.collective == "native"
                else pa.TorchDistributedComm(overlap=D.overlap, shard=shard))
    D.beat()
    f = pa.LeastSquares(A, b, comm=comm)
    # row blocks as a row TEAM (one read of A per iteration; csrc/pg_gemv_tn4.hip): the ranks map each other's inbox through
    # IPC handles.  The context outlives this record, so the mode is switched off again for every other layout.
    teams = bool(row_teams) and layout == "rows" and D.world > 1
    # (ranks SHARING one device -- the one-GPU tests -- must all be resident together: each takes its share of the compute units)
    wgs = -D.world if (teams and getattr(D, "share_device", False)) else 0  # (-k: the default number of workgroups divided by k)
    pa.attach_row_team(ctx, *((None, None) if teams else (1, 0)), max_workgroups=wgs)
    zero_n = pa.HIPVector.zeros(n_loc, dtype, ctx)
    _, g0 = f.value_and_gradient(zero_n)  # = -A'b (row blocks: all-reduced; column blocks: this rank's columns)
    g0_inf = float(g0.norm_inf())
    if cols:
        g0_inf = D.reduce_scalar(g0_inf, dist.ReduceOp.MAX)
    lam = dtype(0.1) * dtype(g0_inf)
    Lf = None
    if mode == "fixed":
        f0 = pa.LeastSquares(A, pa.HIPVector.zeros(m_loc, dtype, ctx), comm=comm)  # x -> A'A x
        v = pa.HIPVector.zeros(n_loc, dtype, ctx).fill_(1.0 / math.sqrt(n))
        w = v.similar()
        nrm = dtype(1)
        for _ in range(30):
            D.beat()
            f0.value_and_gradient(v, out=w)
            nrm2 = float(w.norm()) ** 2
            if cols:
                nrm2 = D.reduce_scalar(nrm2, dist.ReduceOp.SUM)
            nrm = dtype(math.sqrt(nrm2))
            v.axpby_(1.0 / float(nrm), w)
        Lf = dtype(1.1) * nrm  # ||A||^2 estimate (+10 % margin: power iteration under-estimates)
        del f0
    ctx.sync()
    return {"A": A, "b": b, "f": f, "comm": comm, "lam": lam, "Lf": Lf, "zero_n": zero_n, "m_glob": m_glob, "n": n,
            "m_loc": m_loc, "n_loc": n_loc, "layout": layout, "dtype": dtype, "setup_s": time.perf_counter() - t0, "seed": seed,
            "row_teams": teams}


def run_ffb(pa, ctx, D, P, mode, sweeps, steps, warmup, kernel_events, workload_name=None, scaling="strong", sustain=0.0):
    """W untimed + K timed FastForwardBackward iterations on a prepared problem; returns the record (every rank) with
    value = K / max-over-ranks(elapsed), the HIP-event roofline of the dominant sweep kernel and the problem's config."""
    import numpy as np

    dtype, n, m_glob, m_loc, n_loc, layout = P["dtype"], P["n"], P["m_glob"], P["m_loc"], P["n_loc"], P["layout"]
    cols = layout == "cols"
    es = np.dtype(dtype).itemsize
    Lf = P["Lf"] if mode == "fixed" else None
    iteration = pa.FastForwardBackwardIteration(f=P["f"], g=pa.NormL1(P["lam"]), x0=P["zero_n"], Lf=Lf,
                                                single_sweep=sweeps == "one")
    it = iter(iteration)
    state = next(it)  # init (k = 1)
    D.beat()
    stop_rule = lambda s: float(s.res_inf) / float(s.gamma) <= 1e-6  # benchmarks.jl:57 (evaluated, not acted on)
    for _ in range(warmup):
        state = next(it)
        stop_rule(state)
        D.beat()
    passes0 = iteration.counters.get("a_passes", 0)
    comm = P["comm"]
    calls0, elems0 = (getattr(comm, "calls", 0), getattr(comm, "elements", 0)) if comm is not None else (0, 0)
    team_stats0 = pa.row_team_stats(ctx) if P.get("row_teams") else None  # (the telemetry counts since pg_ctx_set_row_team: the record reports ITS share)
    ctx.profile(kernel_events != "none", kernels=None if kernel_events == "all" else ("gemv_n_partial", "gemv_t", "gemv_tn"))
    ctx.profile_reset()
    D.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        state = next(it)
        stop_rule(state)
    D.barrier()
    elapsed = time.perf_counter() - t0
    D.beat()
    prof = ctx.profile_read()
    ctx.profile(False)
    a_passes = iteration.counters.get("a_passes", 0) - passes0
    sustained = None
    if sustain > 0:  # the same iteration kept going for `sustain` seconds of wall clock (no event pairs): a window long enough
        D.barrier()   # for an outside sampler to see, and a check that the K-step figure is not a burst
        t1, k1 = time.perf_counter(), 0
        while time.perf_counter() - t1 < sustain:
            for _ in range(10):
                state = next(it)
                stop_rule(state)
            k1 += 10
            D.beat()
        D.barrier()
        dt1 = time.perf_counter() - t1
        sustained = {"seconds": round(dt1, 2), "steps": k1, "value": round(k1 / dt1, 4), "ms_per_step": round(1e3 * dt1 / k1, 4)}
    fallbacks = int(iteration.counters.get("sweep_fallbacks", 0))  # steps redone with two sweeps (team-sweep timeout / refusal)
    in_library = None
    if mode == "fixed" and D.world == 1 and getattr(iteration, "_fused", None) is not None:
        # the same K iterations enqueued by the library without a host round trip in between (pg_iter_run_batched: with a
        # fixed step nothing the host decides is needed per iteration; the stopping rule is then looked at once per batch).
        # Reported beside `value`, which stays the stepped loop of the reference's iterator (one read-back per iteration).
        try:
            ctx.sync()
            t2 = time.perf_counter()
            k2, _ = iteration._fused.run(0, steps, 0.0, check_every=steps)
            ctx.sync()
            dt2 = time.perf_counter() - t2
            in_library = {"steps": int(k2), "check_every": steps, "value": round(k2 / dt2, 4), "ms_per_step": round(1e3 * dt2 / max(k2, 1), 4)}
        except Exception as e:  # this side measurement must never cost the record measured above
            in_library = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        D.beat()
    elapsed = D.max_over_ranks(elapsed)
    its = steps / elapsed
    sweeps_done = a_passes / max(steps, 1)  # reads of A per iteration actually executed
    # the algorithmic figure counts the passes the mode REQUIRES when A x and A' r are separate sweeps (2 for
    # fixed-step FB / FFB and for adaptive FFB with the residual pair); the single-sweep iteration moves fewer bytes --
    # both are reported, labelled
    passes_alg = max(2.0, sweeps_done) if sweeps == "two" or layout == "rows" else 2.0
    bytes_iter_local = passes_alg * m_loc * n_loc * es + 10 * n_loc * es + 3 * m_loc * es
    bytes_moved_local = sweeps_done * m_loc * n_loc * es + 10 * n_loc * es + 3 * m_loc * es
    kern = {}
    n_cnt = prof["gemv_n_partial"][0]
    for name, vec_bytes in (("gemv_n_partial", n_loc * es), ("gemv_t", m_loc * es + n_loc * es),
                            ("gemv_tn", (m_loc + 7 * n_loc) * es)):
        cnt, ms = prof[name]
        if cnt:
            avg_ms = ms / cnt
            # with a collective attached pass T may run as several column-chunk launches per evaluation
            evals = max(a_passes - n_cnt - prof["gemv_tn"][0], 1) if name == "gemv_t" else cnt
            launch_bytes = (m_loc * n_loc * es + vec_bytes) * evals / cnt
            kern[name] = {"launches": cnt, "avg_ms": avg_ms, "bytes": launch_bytes, "launches_per_pass": cnt / evals,
                          "GBps": launch_bytes / (avg_ms * 1e-3) / 1e9}
    dom = max(kern, key=lambda k_: kern[k_]["avg_ms"]) if kern else None
    roofline = None
    if dom:
        traffic, traffic_src, stale = (None, None, False)
        if workload_name and D.world == 1 and layout == "none" and dtype == np.float32:
            traffic, traffic_src, stale = pmc_traffic(workload_name, dom)
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(kern[dom]["GBps"], 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(kern[dom]["GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_source": traffic_src, "traffic_stale": stale,
                    "avg_launch_ms": round(kern[dom]["avg_ms"], 4), "launches": kern[dom]["launches"],
                    "algorithmic_bytes_per_launch": int(kern[dom]["bytes"]),
                    "per_kernel": {k_: {"avg_ms": round(v["avg_ms"], 4), "GBps": round(v["GBps"], 1), "launches": v["launches"],
                                        "launches_per_pass": round(v["launches_per_pass"], 2)} for k_, v in kern.items()},
                    "whole_iteration": {"algorithmic_bytes_per_gpu": int(bytes_iter_local),
                                        "GBps_per_gpu": round(bytes_iter_local * its / 1e9, 1),
                                        "frac": round(bytes_iter_local * its / 1e9 / HBM_PEAK_GBS, 4),
                                        "sweeps_of_A_per_iteration": round(sweeps_done, 3),
                                        "hbm_bytes_moved_per_gpu": int(bytes_moved_local),
                                        "hbm_GBps_moved_per_gpu": round(bytes_moved_local * its / 1e9, 1),
                                        "frac_of_bytes_moved": round(bytes_moved_local * its / 1e9 / HBM_PEAK_GBS, 4),
                                        "note": "algorithmic_bytes = SURVEY 8(d): A x and A' r as separate passes (2 m n s "
                                                "+ vectors); frac > 1 means the iteration moves fewer bytes than that (the "
                                                "single sweep reads A once); *_moved = bytes actually read/written"}}
    dname = "Float32" if dtype == np.float32 else "Float64"
    rec = {
        "value": round(its, 4), "unit": "it/s", "steps": steps, "warmup": warmup, "ms_per_step": round(1e3 * elapsed / steps, 4),
        "scaling": scaling,
        # The driver's record keeps the first ~22 SCALAR keys of `config` and drops nested values: what identifies the record comes
        # first, everything else about the problem sits under `problem` (kept in profiles/, dropped by the driver)
        "config": {"workload": "FFB LASSO m=%d n=%d %s, %s step, %s" % (
            m_glob, n, dname, mode, "one GPU" if layout == "none" else
            "%s of A sharded over %d GPU(s)" % ("columns" if cols else "rows", D.world)),
            "m": m_glob, "n": n, "mode": mode, "sharding": layout, "shards": D.world if layout != "none" else 1,
            "a_passes_per_step": a_passes / max(steps, 1), "sweep_fallbacks": fallbacks,
            "sweeps": sweeps if (layout != "rows" or P.get("row_teams")) else "two", "row_teams": bool(P.get("row_teams")),
            "problem": {"m_per_gpu": m_loc, "n_per_gpu": n_loc, "lambda": float(P["lam"]), "Lf": float(Lf) if Lf is not None else None,
                        "seed": P["seed"], "setup_s": round(P["setup_s"], 2)},
            "final": {"gamma": float(state.gamma), "f_x": float(state.f_x), "g_z": float(state.g_z),
                      "res_inf_over_gamma": float(state.res_inf) / float(state.gamma)}},
        "roofline": roofline,
    }
    if P.get("row_teams"):  # how the granule exchange went (sweeps, waves that had to wait, polls spent waiting), this rank
        rec["config"]["row_team_stats"] = {k_: v - team_stats0.get(k_, 0) for k_, v in pa.row_team_stats(ctx).items()}  # the timed steps (and the sustain window)
        # every knob that was in force, as the library reports it for its last sweep ("W=1 U=8 C=2 LAG=2 LAGR=2 PF=2 WGS=4 K1=1 PAIR=0
        # SPIN=2097152 WG=1024"), and the share of wave-steps that did not find their granules at the first look
        geom_text, geom = pa.row_team_geometry(ctx)
        late = late_fraction(rec["config"]["row_team_stats"], geom, n)
        rec["config"]["row_team_stats"]["late_fraction"] = late
        rec["config"]["row_team_geometry"] = geom_text + ("" if late is None else " late=%.4f" % late)  # (one scalar: the driver's record keeps ~20)
        rec["config"]["row_team_selftest"] = getattr(ctx, "_row_team_selftest", None)  # the scalar exchange tried at attach time
        if D.world > 1:  # ... and whether EVERY rank's came back right (what an upgrade of the top-level record asks)
            import torch.distributed as dist

            rec["config"]["row_team_selftest_all_ranks"] = bool(D.reduce_scalar(1.0 if rec["config"]["row_team_selftest"] == "ok" else 0.0,
                                                                                dist.ReduceOp.MIN) > 0.5)
    if layout != "none":
        calls = getattr(comm, "calls", 0) - calls0
        elems = getattr(comm, "elements", 0) - elems0
        rec["ranks_seen_by_rccl"] = D.ranks_seen()
        rec["collective"] = {"backend": "rccl" if D.backend == "nccl" else D.backend, "through": D.collective,
                             "allreduce_calls_per_step": round(calls / max(steps, 1), 3) if calls else None,
                             "allreduce_payload_bytes_per_call": int(elems / calls * es) if calls else None,
                             "layout_payload": ("[A v partial (m) ; 8 N scalar slots]" if cols else
                                                "none in the steady state: per-column granules through the peers' inboxes"
                                                if P.get("row_teams") else "[grad (n) ; f]")}
    del it, iteration
    if sustained is not None:
        rec["sustained"] = sustained
    if in_library is not None:
        rec["in_library_loop"] = in_library
    return rec


# ---------------------------------------------------------------------------------------------------------------
# BASELINE configs 3 and 4 (N = 1 `also` records)
# ---------------------------------------------------------------------------------------------------------------
def run_config3(pa, ctx, n=10_000_000, steps=200, beat=lambda: None):
    """DouglasRachford on a box-constrained QP with diagonal Hessian (douglas_rachford.jl:53-70), n = 10^7, Float32:
    stepping from the host (one fused sweep per iteration) and the in-library loop (64 iterations per sweep, two sweeps in flight)."""
    import numpy as np

    dtype = np.float32
    rng = np.random.default_rng(0)
    d = (0.1 + rng.random(n, dtype=np.float32)).astype(dtype)
    q = rng.standard_normal(n, dtype=np.float32)
    x0 = np.zeros(n, dtype)
    lo, hi, gamma = dtype(-0.5), dtype(0.25), dtype(1.0)
    out = {"label": "config3", "unit": "it/s",
           "config": {"workload": "DouglasRachford box-constrained QP n=%d Float32 (SeparableQuadratic + IndBox)" % n}}
    it = iter(pa.DouglasRachfordIteration(f=pa.SeparableQuadratic(d, q), g=pa.IndBox(lo, hi), x0=x0, gamma=gamma,
                                          materialize=False))
    for _ in range(20):
        s = next(it)
    beat()
    ctx.profile(True)
    ctx.profile_reset()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        s = next(it)
        float(s.res_inf) / float(gamma) <= 1e-8  # the stop rule, evaluated every iteration like the driver loop
    ctx.sync()
    dt = time.perf_counter() - t0
    beat()
    cnt, ms = ctx.profile_read()["dr_step"]
    ctx.profile(False)
    b5 = 5 * n * 4  # x, d, q in; x, y out
    out["stepping"] = {"value": round(steps / dt, 1), "ms_per_step": round(1e3 * dt / steps, 5), "steps": steps,
                       "roofline": {"bound": "hbm", "kernel": "dr_step", "avg_launch_ms": round(ms / cnt, 5),
                                    "algorithmic_bytes_per_launch": b5, "achieved": round(b5 / (ms / cnt * 1e-3) / 1e9, 1),
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(b5 / (ms / cnt * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
    out["stepping"]["lookahead"] = True  # iteration k + 1 is in flight while the host reads iteration k's scalar (pg_dr_step_async)
    try:  # the same loop with one launch and one read-back per iteration (rounds 1-3), for the difference
        it1 = iter(pa.DouglasRachfordIteration(f=pa.SeparableQuadratic(d, q), g=pa.IndBox(lo, hi), x0=x0, gamma=gamma,
                                               materialize=False, lookahead=False))
        for _ in range(20):
            s1 = next(it1)
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(steps):
            s1 = next(it1)
            float(s1.res_inf) / float(gamma) <= 1e-8
        ctx.sync()
        out["stepping"]["no_lookahead_it_s"] = round(steps / (time.perf_counter() - t1), 1)
        del it1, s1
    except Exception as e:  # noqa: BLE001
        out["stepping"]["no_lookahead_it_s"] = None
    beat()
    # The same kernel WITHOUT a marker packet around every launch: 100 launches back to back (no scalar read-back) between
    # ONE event pair.  The per-launch pairs above put two marker packets next to a ~35 us kernel and read ~3 us more than
    # rocprofv3's kernel-trace does for the same launches; this figure is the one that agrees with the profiler.
    try:
        import ctypes as C

        import torch

        from proximalalgorithms.jl_amd._lib import call

        fq, gb = pa.SeparableQuadratic(d, q), pa.IndBox(lo, hi)
        xs = pa.HIPVector.from_numpy(x0, ctx)
        ys = xs.similar()
        dv, dsc, qv, qsc = fq.c_params()
        p0, p1 = gb.g_params()
        stream = torch.cuda.ExternalStream(ctx.stream) if ctx.stream else torch.cuda.current_stream()
        step = lambda: call("pg_dr_step", ctx.handle, xs.pg_dtype, xs.n, xs.vp, ys.vp, None, None, None, dv, dsc, qv, qsc, gb.g_kind,
                            p0, p1, float(gamma), None)
        for _ in range(10):
            step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(100):
            step()
        e1.record(stream)
        e1.synchronize()
        b2b = e0.elapsed_time(e1) / 100
        r = out["stepping"]["roofline"]
        r["back_to_back_ms"] = round(b2b, 5)
        r["back_to_back_frac"] = round(b5 / (b2b * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        # the record's roofline is the figure that agrees with the profiler; the per-launch pairs stay beside it
        r["event_pair_ms"], r["event_pair_frac"] = r["avg_launch_ms"], r["frac"]
        r["avg_launch_ms"] = r["back_to_back_ms"]
        r["achieved"] = round(b5 / (b2b * 1e-3) / 1e9, 1)
        r["frac"] = r["back_to_back_frac"]
        r["note"] = ("avg_launch_ms / frac: 100 launches between ONE HIP event pair = what rocprofv3 --kernel-trace reports for the kernel "
                     "(profiles/r3_dr_counters.md); event_pair_ms: an event pair around EVERY launch of the stepped loop (two marker packets "
                     "per ~35 us kernel read ~3 us more).  The kernel is a copy-like stream (3 n-vectors "
                     "in, 2 out): its ceiling is the device's read+write rate (5.5-6.0 TB/s = 0.69-0.75 of the 8 TB/s read peak, "
                     "profiles/r2_stream_ceiling.log), and the 200 MB working set gains nothing from sitting in the 256 MiB Infinity "
                     "Cache, which streams at the HBM rate (profiles/r3_mall_panel.md)")
        del xs, ys
    except Exception as e:  # a side measurement must not cost the record
        out["stepping"]["roofline"]["back_to_back_ms"] = None
        out["stepping"]["roofline"]["note"] = "back-to-back timing failed: %s" % str(e)[:120]
    block = 64
    nst = max(steps, 20 * block) // block * block
    itn = pa.DouglasRachfordIteration(f=pa.SeparableQuadratic(d, q), g=pa.IndBox(lo, hi), x0=x0, gamma=gamma, materialize=False)
    itn.device_run(2 * block, 0.0, block)
    ctx.sync()
    t0 = time.perf_counter()  # timed without event pairs: two sweeps are in flight and a marker packet between them costs ~5 us
    s, k = itn.device_run(nst, 0.0, block)
    ctx.sync()
    dt = time.perf_counter() - t0
    ctx.profile(True)
    ctx.profile_reset()
    itn.device_run(4 * block, 0.0, block)  # the kernel's own duration, from a second (bracketed) run
    ctx.sync()
    cnt, ms = ctx.profile_read()["dr_step"]
    ctx.profile(False)
    # the bound that applies to the blocked kernel: VALU issue.  Per 4 elements and iteration 18 v_pk_*_f32 (half rate on
    # gfx950: 2 issue slots each), 4 v_med3 and 2 v_max3 = 42 slots = 10.5 per element; a CU issues 64 lane-slots per clock
    info = ctx.device_info()
    slots = 10.5
    valu_floor_ms = block * n * slots / (info["compute_units"] * 64.0 * info["clock_khz"] * 1e3) * 1e3
    out["device_loop"] = {"value": round(nst / dt, 1), "ms_per_step": round(1e3 * dt / nst, 6), "steps": nst,
                          "iterations_per_launch": block,
                          "valu": {"issue_slots_per_element_iteration": slots, "compute_units": info["compute_units"],
                                   "clock_khz": info["clock_khz"], "floor_ms_per_launch": round(valu_floor_ms, 5),
                                   "frac": round(valu_floor_ms / (ms / cnt), 4),
                                   "note": "floor = the K iterations' VALU issue time alone (loads, stores and the reduction excluded)"},
                          "roofline": {"bound": "valu (the prox's division, evaluated exactly) / hbm", "kernel": "dr_block<%d>" % block,
                                       "avg_launch_ms": round(ms / cnt, 5), "algorithmic_bytes_per_launch": b5,
                                       "achieved": round(b5 / (ms / cnt * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(b5 / (ms / cnt * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
    out["value"] = out["device_loop"]["value"]
    out["ms_per_step"] = out["device_loop"]["ms_per_step"]
    out["steps"] = nst
    out["roofline"] = out["stepping"]["roofline"]
    return out


def config4_problem(pa, ctx, m=16384, n=1_000_000):
    """BASELINE config 4's instance: logistic loss + L1 on the synthetic 16384 x 10^6 Float32 matrix (resident once for the family)"""
    import numpy as np

    dtype = np.float32
    A = pa.HIPMatrix.synthetic(m, n, dtype, seed=0, ctx=ctx)
    rng = np.random.default_rng(12345)
    k = max(1, n // 1000)
    x_true = np.zeros(n, dtype)
    x_true[rng.choice(n, size=k, replace=False)] = rng.standard_normal(k).astype(dtype)
    b = A.mul(pa.HIPVector.from_numpy(x_true, ctx))
    b.axpby_(1.0, b, 0.01, pa.HIPVector.from_numpy(rng.standard_normal(m).astype(dtype), ctx))
    f = pa.LogisticLoss(b)
    _, g0 = f.value_and_gradient(pa.HIPVector.zeros(m, dtype, ctx))
    lam = dtype(0.1) * A.mul_adjoint(g0).norm_inf()
    return {"A": A, "f": f, "lam": lam, "m": m, "n": n}


def run_config4(pa, ctx, prob, algo="PANOC", steps=20, warmup=3, beat=lambda: None):
    """PANOC (panoc.jl:138-255), ZeroFPR (zerofpr.jl:142-220) or PANOCplus (panocplus.jl:168-240) -- L-BFGS memory 5, adaptive step -- on
    config 4's instance.  PANOC is BASELINE config 4 itself (K timed iterations after W warm-up steps); the other two are its family
    (SURVEY 8(f) row 4), timed over their FIRST iterations (no warm-up: that is where their line searches backtrack)."""
    import numpy as np

    A, f, lam, m, n = prob["A"], prob["f"], prob["lam"], prob["m"], prob["n"]
    dtype = np.float32
    iteration = getattr(pa, algo + "Iteration")(f=f, A=A, g=pa.NormL1(lam), x0=np.zeros(n, dtype))
    it = iter(iteration)
    beat()
    s = next(it)
    for _ in range(warmup):
        s = next(it)
        beat()
    p0 = iteration.counters.get("A_passes", 0)
    ctx.profile(True)
    ctx.profile_reset()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        s = next(it)
        float(s.res.norm_inf()) / float(s.gamma) <= 1e-8
        beat()
    ctx.sync()
    dt = time.perf_counter() - t0
    prof = ctx.profile_read()
    ctx.profile(False)
    passes = iteration.counters.get("A_passes", 0) - p0
    per = {k_: {"launches": prof[k_][0], "avg_ms": round(prof[k_][1] / prof[k_][0], 4),
                "GBps": round(m * n * 4 / (prof[k_][1] / prof[k_][0] * 1e-3) / 1e9, 1)}
           for k_ in ("gemv_n_partial", "gemv_t", "gemv_tn") if prof[k_][0]}
    dom = max(per, key=lambda k_: per[k_]["avg_ms"] * per[k_]["launches"])
    gemv_ms = sum(prof[k_][1] for k_ in per)
    return {"label": "config4" if algo == "PANOC" else "config4_" + algo.lower(), "value": round(steps / dt, 3), "unit": "it/s", "steps": steps,
            "warmup": warmup, "ms_per_step": round(1e3 * dt / steps, 4),
            "config": {"workload": "%s logistic + L1, m=%d n=%d Float32, LBFGS(5), adaptive step" % (algo, m, n),
                       "A_passes_per_step": round(passes / steps, 4), "lambda": float(lam),
                       "final": {"gamma": float(s.gamma), "res_inf_over_gamma": float(s.res.norm_inf()) / float(s.gamma)}},
            "roofline": {"bound": "hbm", "kernel": dom, "avg_launch_ms": per[dom]["avg_ms"],
                         "algorithmic_bytes_per_launch": m * n * 4, "achieved": per[dom]["GBps"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(per[dom]["GBps"] / HBM_PEAK_GBS, 4), "per_kernel": per,
                         "gemv_time_fraction_of_step": round(gemv_ms * 1e-3 / dt, 4)}}


# ---------------------------------------------------------------------------------------------------------------
def summary_row(r):
    """[it/s, ms_per_step, dominant kernel, roofline frac] of a record -- the compact form nested under `config`"""
    if not isinstance(r, dict) or "value" not in r:
        return {"error": (r or {}).get("error", "not measured")[:120]} if isinstance(r, dict) else None
    roof = r.get("roofline") or {}
    return [r.get("value"), r.get("ms_per_step"), roof.get("kernel"), roof.get("frac")]


def wall_ledger(d):
    """{stage: seconds} of a finished line: import, init, the top-level record and every further record (their set-up, warm-up,
    timed steps, profile read-back and settling waits included), and what is left over (finalize, interpreter start)"""
    w = dict((d.get("job") or {}).get("wall") or {})
    out = {"import": w.get("import_s"), "init": w.get("init_s"), "main": d.get("wall_s")}
    for k_, v in d.items():
        if isinstance(v, dict) and "wall_s" in v:
            out[k_] = v["wall_s"]
    known = sum(v for v in out.values() if v)
    if w.get("total_s") is not None:
        out["other"] = round(max(0.0, w["total_s"] - known), 2)
        out["total"] = w["total_s"]
    return out


def extrapolate_ledger(d, m_full, n_full, steps_full=50, warmup_full=5, gen_rate=2.4e12, stream_rate=7.0e12, clear_rate=30e9):
    """The ledger of a reduced-size dry run (`--gpus N --share-device --backend gloo --m .. --n ..`) scaled to the full
    problem: every record keeps its measured wall time (process start, collective set-up, Python, launch overheads -- what a
    dry run CAN measure) and gains what the full-size block adds: its generation at the measured 2.4 TB/s, the streaming
    passes of set-up and iterations at 7 TB/s, and the driver's background clearing of the block freed before it (30 GB/s,
    capped at 6 s like bench.py's settle).  Returns ({stage: seconds}, total)."""
    led = wall_ledger(d)
    world = d.get("n_gpus", 1)
    es = 4 if d.get("dtype", "f32") == "f32" else 8
    out = dict(led)

    def add(key, rec, steps, warm):
        cfg = rec.get("config") or {}
        if "m" not in cfg:
            return
        weak = rec.get("scaling") == "weak"
        m_glob = m_full * world if weak else m_full
        blk_full = m_glob * n_full * es / world  # bytes of this rank's block at full size
        blk_dry = cfg["m"] * cfg["n"] * es / world
        passes = float(cfg.get("a_passes_per_step") or 2.0)
        extra_bytes = max(0.0, blk_full - blk_dry)
        setup = extra_bytes / gen_rate + 31 * 2 * extra_bytes / stream_rate  # generation, 30 power iterations + one gradient
        timed = (steps + warm + 2) * passes * extra_bytes / stream_rate
        settle = min(6.0, blk_full / clear_rate + 0.3)
        out[key] = round((led.get(key) or 0.0) + setup + timed + settle, 2)

    add("main", d, steps_full, warmup_full)
    for k_, v in d.items():
        if isinstance(v, dict) and "wall_s" in v and "config" in v:
            add(k_, v, max(4, min(steps_full, 20)), 3)
    total = sum(v for k_, v in out.items() if k_ not in ("total", "other") and v) + (led.get("other") or 0.0)
    out["total"] = round(total, 2)
    return out, total


SHORT_LABELS = {"headline_adaptive": "ad", "config2": "c2", "config3": "c3", "config4": "c4", "config4_zerofpr": "zf", "config4_panocplus": "pp",
                "config5_column_block": "c5", "headline_row_block_n8": "r8", "rows_2proc_two_sweeps": "r2",
                "rows_2proc_row_team": "rt", "rows_two_sweeps": "2s", "cols_strong": "co", "rows_strong": "ro",
                "config5_weak_rows": "5r", "config5_weak_cols": "5c", "rows_strong_teams": "tm",
                "config5_weak_rows_teams": "5t"}
SUMMARY_BUDGET = 120  # characters: the driver's BENCH_rNN.json cuts a scalar string of `config` at about 128


def summary_string(records):
    """Every further record of the line in ONE scalar string that FITS the driver's record (it keeps about twenty scalar keys of
    `config`, nothing nested, and cuts a string at about 128 characters): `label=it/s@frac[/reads-of-A]` joined by `;`, two-letter
    labels (SHORT_LABELS), three significant digits (`k` = thousands), the fraction of the roofline in PER CENT, the reads of A per
    iteration only where they are not 1 (config 3: `c3=<in-library it/s>|<stepped it/s>@percent`; a failed record: `label=!reason`).
    The full-precision figures stay in the line itself (`also`, `also_summary`); parse_summary_string is the inverse, to the
    digits kept."""
    parts = []
    for label, r in records:
        key = SHORT_LABELS.get(label, label)
        if not isinstance(r, dict) or "value" not in r:
            why = str((r or {}).get("error", "not measured")) if isinstance(r, dict) else "not measured"
            parts.append("%s=!%s" % (key, "".join(ch if ch not in ";=@/()|" else " " for ch in why)[:24].strip()))
            continue
        roof, cfg = r.get("roofline") or {}, r.get("config") or {}
        passes = cfg.get("a_passes_per_step", cfg.get("A_passes_per_step"))
        if isinstance(r.get("stepping"), dict):
            st = r["stepping"]
            txt = "%s|%s@%s" % (_v3(r["value"]), _v3(st.get("value") or 0.0, 2), _f2((st.get("roofline") or {}).get("frac")))
        else:
            txt = "%s@%s" % (_v3(r["value"]), _f2(roof.get("frac")))
            if passes is not None and abs(float(passes) - 1.0) >= 0.05:
                txt += "/%.2g" % float(passes)
        parts.append("%s=%s" % (key, txt))
    return ";".join(parts)


def _v3(v, digits=3):
    v = float(v)
    return ("%.*gk" % (digits, v / 1e3)) if v >= 999.5 else ("%.*g" % (digits, v))


def _f2(v):
    return "?" if v is None else "%d" % round(100.0 * float(v))


def _unf2(t):
    """per cent of the short form ("73") or the fraction of rounds 4-5's long form ("0.733")"""
    if t in ("", "?"):
        return None
    return float(t) if "." in t else float(t) / 100.0


def _unv3(t):
    return float(t[:-1]) * 1e3 if t.endswith("k") else float(t)


def parse_summary_string(text):
    """{label: {"it_s", "frac", "a_passes"} | {"it_s", "stepping_it_s", "stepping_frac"} | {"error"}} of a summary_string (reads of A
    that the string leaves out are 1); also reads the long form of rounds 4-5 (`cfg3=3e5(step 2e4@0.73)`)."""
    out = {}
    for part in filter(None, text.split(";")):
        key, _, val = part.partition("=")
        if val.startswith("!"):
            out[key] = {"error": val[1:]}
        elif "(step " in val or "|" in val:
            if "|" in val:
                head, _, rest = val.partition("|")
            else:
                head, _, rest = val.partition("(step ")
            st, _, fr = rest.rstrip(")").partition("@")
            out[key] = {"it_s": _unv3(head), "stepping_it_s": _unv3(st), "stepping_frac": _unf2(fr)}
        else:
            its, _, rest = val.partition("@")
            fr, _, ps = rest.partition("/")
            out[key] = {"it_s": _unv3(its), "frac": _unf2(fr), "a_passes": 1.0 if ps == "" else (None if ps == "?" else float(ps))}
    return out


class Job:
    """what rank 0 needs to print the line at any moment: the records measured so far"""

    def __init__(self, args, world, rank):
        self.args, self.world, self.rank = args, world, rank
        self.main_rec, self.extra, self.cpu, self.meta = None, {}, None, {}
        self.layout_note = {}  # N > 1, row blocks on top: {"row_layout": "row_teams" | "two_sweeps", "row_layout_reason": ...}
        self.json_fd = None

    def line(self, error=None, stage=None):
        args = self.args
        if self.main_rec is None:
            d = error_line(args, error or "not measured", stage)
            d["metric"] = metric_name(args, self.world)
            d["n_gpus"] = self.world
        else:
            r = self.main_rec
            src = dict(r["config"])
            # scalars first, in the order of what a reader of the driver's record needs; nested values (dropped there) last
            config = {k_: v for k_, v in src.items() if not isinstance(v, (dict, list))}
            config.update(self.layout_note)
            also = self.extra.get("also")
            if also is not None:
                config["also"] = summary_string([(a.get("label", "?"), a) for a in also])
            subs = [(k_, v) for k_, v in self.extra.items() if k_ != "also" and isinstance(v, dict) and ("value" in v or "error" in v)]
            if subs:
                config["layouts"] = summary_string(subs)
            if "sustained" in r:
                config["sustained_it_s"] = r["sustained"]["value"]
            if "in_library_loop" in r and "value" in r["in_library_loop"]:
                config["in_library_loop_it_s"] = r["in_library_loop"]["value"]
            config.update({k_: v for k_, v in src.items() if isinstance(v, (dict, list))})
            if also is not None:
                config["also_summary"] = {a.get("label", "?"): summary_row(a) for a in also}
            if subs:
                config["layouts_summary"] = {k_: summary_row(v) for k_, v in subs}
            d = {"metric": metric_name(args, self.world), "value": r["value"], "unit": "it/s", "n_gpus": self.world,
                 "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"], "higher_is_better": True,
                 "scaling": args.scaling, "vs_baseline": None, "dtype": args.dtype, "data": "synthetic", "config": config,
                 "roofline": r["roofline"], "cpu_baseline": self.cpu}
            for k_ in ("ranks_seen_by_rccl", "collective", "sustained", "in_library_loop", "wall_s"):
                if k_ in r:
                    d[k_] = r[k_]
            if error is not None:
                d["error"], d["stage"] = error, stage
        d["job"] = self.meta
        d.update(self.extra)
        return d

    def write(self, error=None, stage=None):
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.write(self.json_fd, (json.dumps(self.line(error, stage)) + "\n").encode())


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        sys.exit(self_launch(args))

    # SIGTERM is taken by the watchdog's sigwait thread: block it here, before any library starts a thread of its own
    signal.pthread_sigmask(signal.SIG_BLOCK, {signal.SIGTERM})
    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
    # stdout carries exactly ONE line (the JSON, written last by rank 0): RCCL prints a version banner through C
    # stdio on fd 1 when a communicator is created, so everything else on fd 1 is routed to stderr
    sys.stdout.flush()
    job = Job(args, world, rank)
    job.json_fd = os.dup(1)
    os.dup2(2, 1)
    wd = Watchdog(rank, args.stall_timeout, job.write, inject=args.inject_fault)
    try:
        rc = run_rank(args, job, wd, world, rank, local_rank)
    except BaseException as e:  # a rank that raises still leaves a line (rank 0) and a non-zero exit code
        if isinstance(e, SystemExit) and not e.code:
            raise
        traceback.print_exc()
        wd.close()
        if rank == 0:
            job.write("%s: %s" % (type(e).__name__, str(e)[:400]), wd.stage)
        sys.stderr.flush()
        os._exit(1)  # not sys.exit: a peer stuck in a collective must not keep this process in an atexit handler
    os.close(job.json_fd)
    if rc:
        sys.exit(rc)


def run_rank(args, job, wd, world, rank, local_rank):
    t_rank0 = time.perf_counter()
    wd.enter("import", 600.0, stall=False)  # the first `import torch` on a fresh box pages the image in: minutes, not a hang
    import numpy as np
    import torch
    import torch.distributed as dist

    import proximalalgorithms.jl_amd as pa

    t_imported = time.perf_counter()
    wd.enter("init", args.init_timeout + 30.0, stall=False)
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)  # before the first library call: the context below is created on this device
    job.meta = {"HIP_VISIBLE_DEVICES": os.environ.get("HIP_VISIBLE_DEVICES"), "ROCR_VISIBLE_DEVICES": os.environ.get("ROCR_VISIBLE_DEVICES"),
                "devices_visible": torch.cuda.device_count(), "device": torch.cuda.get_device_name(local_rank),
                "local_rank": local_rank, "backend": None, "rccl_version": None, "collective": None}
    if world > 1 or args.force_comm:
        import datetime

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        if world == 1:
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        tmo = datetime.timedelta(seconds=args.init_timeout)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=tmo)
            try:
                job.meta["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                pass
        else:
            dist.init_process_group("gloo", timeout=tmo)
        job.meta["backend"] = "rccl" if args.backend == "nccl" else args.backend
        job.meta["ranks_seen_by_rccl"] = dist.get_world_size()
        if job.meta["ranks_seen_by_rccl"] != world:  # before any record is timed: a job whose collective spans fewer ranks
            raise RuntimeError("the collective backend reports %d ranks, the job was launched with %d"  # must not print rates
                               % (job.meta["ranks_seen_by_rccl"], world))
    collective = args.collective
    if collective == "auto":  # the library's own RCCL communicator when librccl loads and the job runs on RCCL; else torch.distributed
        collective = "native" if (args.backend == "nccl" and pa.native_rccl_available()) else "torch"
    ctx = pa.get_context(local_rank)
    if collective == "native" and (world > 1 or args.force_comm):
        # First use of the library's own communicator in this job: create it and push one tiny row-sharded gradient through
        # it (every rank holds the row [1 1 1 1]; grad at x = 1 must come back as 4 * world).  Every rank reports success or
        # failure through torch.distributed, and unless ALL succeeded the job continues on the torch collective -- a second
        # communicator that does not come up must not cost the run.  (A hang in here is the watchdog's: stage "init".)
        ok, why = 1, ""
        try:
            comm0 = pa.NativeRcclComm(shard="rows")
            f0 = pa.LeastSquares(np.ones((1, 4), np.float32), np.zeros(1, np.float32), comm=comm0, ctx=ctx)
            _, g0 = f0.value_and_gradient(pa.HIPVector.from_numpy(np.ones(4, np.float32), ctx))
            got = g0.numpy()
            if not np.allclose(got, 4.0 * world):
                ok, why = 0, "self-test all-reduce returned %s, expected %g" % (got.tolist(), 4.0 * world)
            del f0, g0
        except Exception as e:
            ok, why = 0, "%s: %s" % (type(e).__name__, str(e)[:200])
        if world > 1:
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok_all = int(flag.item())
        else:
            ok_all = ok
        if not ok_all:
            sys.stderr.write("bench.py: rank %d: native RCCL communicator unavailable (%s) -> torch.distributed collective\n" % (rank, why or "another rank failed"))
            try:
                from proximalalgorithms.jl_amd._lib import call as _call

                _call("pg_ctx_comm_destroy", ctx.handle)
                ctx._native_comm_shape = None
                ctx.set_column_sharding(0, 0)
            except Exception:
                pass
            collective = "torch"
            job.meta["collective_fallback"] = why or "another rank failed"
    job.meta["collective"] = collective
    D = Dist(world, rank, local_rank, args.backend, collective, args.overlap, args.force_comm)
    D.share_device = bool(args.share_device)
    D.beat = wd.beat

    m_base, n = WORKLOADS[args.workload]
    m_base = args.m or m_base
    n = args.n or n
    m_glob = m_base * world if args.scaling == "weak" else m_base
    dtype = np.float32 if args.dtype == "f32" else np.float64
    # N > 1: ROW blocks are the contract (north_star): in this process group with two sweeps per iteration and
    # the all-reduce of [grad ; f] (n + 1 elements) -- the form that needs nothing but RCCL -- and afterwards, in a process group
    # of its own, as a row team (one read of A per iteration), which becomes the top-level record when it runs clean.  Column
    # blocks (every GPU keeps the single sweep, one all-reduce of m + 8 (N + 1) elements) are a sub-record, labelled as not the
    # contract layout.
    layout = args.sharding
    if layout == "auto":
        layout = "rows"
    if not D.sharded:
        layout = "none"
    named = args.workload if (args.m is None and args.n is None) else None
    if world == 1:  # the shapes further PMC passes were taken on (profiles/pmc_traffic.json)
        named = {(131072, 131072): "long_columns", (2048, 1 << 20): "short_columns"}.get((m_glob, n), named)

    if args.row_teams_child:
        return row_team_child(args, job, wd, pa, ctx, D, world, rank, m_base, n, dtype)
    if args.sustain is None:
        args.sustain = 5.0 if (named == "headline" and world == 1 and not args.force_comm) else 0.0
    # the wall-clock ledger of the job (seconds of this rank): what a first run on N GPUs must fit into --launch-timeout
    job.meta["wall"] = {"import_s": round(t_imported - t_rank0, 2), "init_s": round(time.perf_counter() - t_imported, 2)}
    wd.enter("main", args.record_timeout + args.sustain)
    t_rec = time.perf_counter()
    P = setup_lasso(pa, ctx, D, m_glob, n, dtype, args.seed, layout, args.mode, row_teams=args.row_teams)
    job.main_rec = run_ffb(pa, ctx, D, P, args.mode, args.sweeps, args.steps, args.warmup, args.kernel_events,
                           workload_name=named, scaling=args.scaling, sustain=args.sustain)
    if P.get("row_teams") and args.mode == "fixed":  # (`--row-teams` in the job's own process group: the same two-geometry rule as the child's)
        job.main_rec = row_team_second_geometry(job.main_rec, lambda: run_ffb(pa, ctx, D, P, args.mode, args.sweeps, args.steps, args.warmup, args.kernel_events,
                                                                               workload_name=named, scaling=args.scaling, sustain=args.sustain), pa, ctx, D, world)
    job.main_rec["wall_s"] = round(time.perf_counter() - t_rec, 2)
    wd.main_done = True
    extra = job.extra
    sub_steps = max(4, min(args.steps, 20))
    if args.no_also:
        pass
    elif world == 1 and not args.force_comm and named == "headline" and args.mode == "fixed":
        also = extra["also"] = []
        # the reference benchmark's own mode: adaptive step (benchmark/benchmarks.jl:55-61), same matrix
        wd.enter("headline_adaptive", args.sub_record_timeout)
        r = run_ffb(pa, ctx, D, P, "adaptive", args.sweeps, sub_steps, 3, args.kernel_events)
        r["label"] = "headline_adaptive"
        also.append(r)
        # the CPU leg needs the headline matrix: take it before the other configs claim the memory
        if rank == 0 and not args.no_cpu_baseline:
            wd.enter("cpu_baseline", 600.0, stall=False)
            job.cpu = cpu_leg(args, P, m_glob, n, np.dtype(dtype).itemsize)
            args.no_cpu_baseline = True
        es = np.dtype(dtype).itemsize
        P = None  # release the 64 GiB matrix
        freed = [m_glob * n * es]
        t_also = time.perf_counter()  # time box of the remaining records (the CPU leg and the settling waits are not part of it)
        settled = [0.0]
        within = lambda: time.perf_counter() - t_also - settled[0] < args.also_budget

        def settle():
            # Freed device memory is cleared by the driver in the background at ~35 GB/s, and kernels that run meanwhile lose
            # 2-4 % (profiles/r3_after_big_free.log: config 2 at 803 it/s for the 2.1 s after a 64 GiB free, 817 from then
            # on).  A record that follows a big free waits that long before its set-up -- untimed, like the set-up itself.
            import gc

            gc.collect()
            ctx.sync()
            wait = min(6.0, freed[0] / 30e9 + 0.3) if args.settle else 0.0
            if freed[0] > (1 << 30) and wait > 0:
                time.sleep(wait)
                settled[0] += wait
            freed[0] = 0

        def also_record(label, fn, frees=0):
            # an extra record that fails (out of memory on a smaller device, a refused shape) must not cost the headline line
            if not within():
                return
            wd.enter(label, args.sub_record_timeout)
            settle()
            try:
                r = fn()
            except (pa.ProxGradError, MemoryError, RuntimeError, OSError, ValueError, subprocess.SubprocessError) as e:
                r = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            r["label"] = label
            also.append(r)
            freed[0] = frees  # bytes of device memory this record gives back when it returns

        def ffb_record(mm, nn, steps_, warm_, key):
            P2 = setup_lasso(pa, ctx, D, mm, nn, dtype, args.seed, "none", "fixed")
            return run_ffb(pa, ctx, D, P2, "fixed", "one", steps_, warm_, args.kernel_events, workload_name=key)

        m2_, n2_ = WORKLOADS["config2"]
        also_record("config2", lambda: ffb_record(m2_, n2_, max(sub_steps, 50), 5, "config2"), frees=m2_ * n2_ * es)
        also_record("config3", lambda: run_config3(pa, ctx, beat=wd.beat))
        # config 4 (PANOC) and its family on ONE resident instance: ZeroFPR (two trial points of its line search per sweep) and
        # PANOCplus (its second pass taken in the next first sweep), each over its first 23 iterations
        c4 = {}

        def config4_record(algo, steps_, warm_):
            if "prob" not in c4:
                c4["prob"] = config4_problem(pa, ctx)
            return run_config4(pa, ctx, c4["prob"], algo, steps=steps_, warmup=warm_, beat=wd.beat)

        also_record("config4", lambda: config4_record("PANOC", 20, 3))
        also_record("config4_zerofpr", lambda: config4_record("ZeroFPR", 23, 0))
        also_record("config4_panocplus", lambda: config4_record("PANOCplus", 23, 0), frees=16384 * 1_000_000 * 4)
        c4.clear()
        # per-GPU block shapes at N = 8, run as problems of their own on one GPU: BASELINE config 5 under the column layout
        # (131072 x 131072: one team sweep per iteration) and the headline under north_star's row layout (2048 x 2^20: the
        # short-column sweep, one wave per column group); the PMC passes of these sweeps were taken on exactly these shapes
        also_record("config5_column_block", lambda: ffb_record(131072, 131072, sub_steps, 3, "long_columns"), frees=131072 * 131072 * es)
        also_record("headline_row_block_n8", lambda: ffb_record(2048, 1 << 20, sub_steps, 3, "short_columns"), frees=2048 * (1 << 20) * es)
        # north_star's ROW layout between two PROCESSES sharing this device (2 x 2048 rows, the headline's N = 8 block length; one
        # process per rank, inboxes mapped through IPC handles): two sweeps + the all-reduce, then the same as a row TEAM (one read
        # of A per iteration, csrc/pg_gemv_tn4.hip).  Each runs as a CHILD job of its own (`bench.py --gpus 2 --share-device ...`).
        if dtype == np.float32:
            # (the child's own limit + the kill margin + the settle wait stay below the stage's deadline)
            t_child = max(30.0, args.sub_record_timeout - 45.0)
            also_record("rows_2proc_two_sweeps", lambda: shared_device_rows_record(4096, 1 << 20, False, sub_steps, wd.beat, timeout=t_child))
            also_record("rows_2proc_row_team", lambda: shared_device_rows_record(4096, 1 << 20, True, sub_steps, wd.beat, timeout=t_child))
        if settled[0] > 0:
            extra["also_settle_s"] = round(settled[0], 2)
    elif world > 1:
        es = np.dtype(dtype).itemsize
        freed = [P["m_loc"] * P["n_loc"] * es]
        P = None
        other = "rows" if layout == "cols" else "cols"

        def extra_record(key, m_rec, lay, scaling, teams=False):
            # an extra record that cannot run (the library refuses the shape on every rank alike) must not cost the line
            wd.enter(key, args.sub_record_timeout)
            t_sub = time.perf_counter()
            if args.settle and freed[0] > (1 << 30):  # the driver clears the block just freed in the background (see `settle` above);
                import gc                             # every rank frees the same number of bytes, so all wait equally long

                gc.collect()
                ctx.sync()
                time.sleep(min(6.0, freed[0] / 30e9 + 0.3))
            freed[0] = (m_rec // world if lay == "rows" else m_rec) * (n if lay == "rows" else -(-n // world)) * es
            try:
                P2 = setup_lasso(pa, ctx, D, m_rec, n, dtype, args.seed, lay, "fixed", row_teams=teams)
                extra[key] = run_ffb(pa, ctx, D, P2, "fixed", "one", sub_steps, 3, args.kernel_events, scaling=scaling)
            except pa.ProxGradError as e:
                # only what every rank sees alike: a refused shape or a failed allocation.  A failed collective or HIP call
                # leaves the ranks out of step -- that ends the job (with the line measured so far)
                if e.code not in (pa.PG_ERR_UNSUPPORTED, pa.PG_ERR_ALLOC):
                    raise
                extra[key] = {"error": str(e)[:300]}
            extra[key]["wall_s"] = round(time.perf_counter() - t_sub, 2)

        if layout == "rows":
            job.layout_note = {"row_layout": "row_teams" if args.row_teams else "two_sweeps",
                               "row_layout_reason": "--row-teams: the top-level record itself ran as a row team" if args.row_teams else
                               "the row-team upgrade has not run yet"}
        if args.scaling == "strong":
            # the same global problem in the other layout (column blocks: not the contract layout)
            extra_record("%s_strong" % other, m_base, other, "strong")
            if other == "cols" and isinstance(extra.get("cols_strong"), dict) and "config" in extra["cols_strong"]:
                extra["cols_strong"]["config"]["note"] = "not the contract layout (north_star prescribes row blocks): reported beside it"
        # BASELINE config 5 and its twins: m_base rows PER GPU (131072 x 2^20 at N = 8), both layouts
        for lay in ("rows", "cols"):
            if args.scaling == "weak" and lay == layout:
                continue
            extra_record("config5_weak_%s" % lay, m_base * world, lay, "weak")
        # north_star's row layout at ONE read of A per iteration (row teams), in a process group of its own: it has never run on
        # real xGMI before the first SCALE collection, and whatever happens in there costs none of the records above.  Its record
        # of the top-level problem, run for the same K steps, REPLACES the two-sweep record on top when it is clean.
        why_not = None
        if args.no_row_teams:
            why_not = "--no-row-teams"
        elif layout != "rows" or args.row_teams:
            why_not = "the top-level record is not the two-sweep row layout"
        elif args.share_device and world > 4:
            # (profiles/r4_row_team_one_gpu.md: five and more processes on one device are time-sliced, their kernels are not all
            # resident, every wave polls -- ~1 it/s; a one-GPU rehearsal artefact, no configuration anyone runs)
            why_not = "more than four rank processes share one device: their sweeps would be time-sliced, not co-resident"
        if why_not is None:
            row_team_records_in_a_child(args, job, wd, ctx, world, rank, freed_bytes=freed[0] if args.settle else 0)
            if rank == 0:
                promote_row_team_record(args, job)
        elif layout == "rows" and not args.row_teams:
            job.layout_note = {"row_layout": "two_sweeps", "row_layout_reason": "row teams not tried: " + why_not}

    if rank == 0 and job.cpu is None and world == 1 and not args.no_cpu_baseline and P is not None:
        wd.enter("cpu_baseline", 600.0, stall=False)
        job.cpu = cpu_leg(args, P, m_glob, n, np.dtype(dtype).itemsize)
    job.meta["wall"]["total_s"] = round(time.perf_counter() - t_rank0, 2)
    wd.enter("finalize", 60.0)
    if world > 1 or args.force_comm:
        try:  # every record is measured: a peer that is already gone must not turn the line into a failure
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("bench.py: rank %d: the closing barrier failed (%s)\n" % (rank, str(e)[:200]))
    wd.close()
    if rank == 0:
        job.write()
    return 0


def shared_device_rows_record(m, n, teams, steps, beat=lambda: None, timeout=170.0):
    """an N = 1 `also` record: `bench.py --gpus 2 --share-device --backend gloo --sharding rows [--row-teams]` as a child job
    (its own launcher, two rank processes on this one device, gloo for the set-up collectives) and what its line says"""
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "2", "--share-device", "--backend", "gloo", "--sharding", "rows", "--m", str(m),
           "--n", str(n), "--steps", str(steps), "--warmup", "3", "--no-also", "--no-cpu-baseline", "--no-row-teams", "--launch-timeout", str(timeout)]
    if teams:
        cmd.append("--row-teams")
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PG_BENCH_ARGV")
           and not k_.startswith("TORCHELASTIC_")}
    beat()
    # Whatever the child does -- no line, a broken line, a hang -- this is an EXTRA record: it returns {"error": ...} and the
    # headline line stands (TimeoutExpired / ValueError / OSError escaped to main() and turned a measured line into a
    # failure; and subprocess.run's timeout killed the launcher only, leaving the rank processes on the GPU).
    try:
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
        try:
            out_s, err_s = proc.communicate(timeout=timeout + 15.0)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, signal.SIGKILL)  # the launcher AND its ranks (one session)
            except ProcessLookupError:
                pass
            out_s, err_s = proc.communicate()
            return {"error": "the two-process child did not finish within %.0f s" % (timeout + 15.0)}
        beat()
        line = _find_line(out_s)
        if line is None:
            return {"error": "the two-process child printed no line (exit code %d)" % proc.returncode, "stderr_tail": err_s.splitlines()[-4:]}
        d = json.loads(line)
    except Exception as e:  # noqa: BLE001
        return {"error": "the two-process child failed: %s: %s" % (type(e).__name__, str(e)[:200])}
    if d.get("value") is None:
        return {"error": "child: %s (stage %s)" % (d.get("error"), d.get("stage"))}
    cfg = d.get("config") or {}
    passes = float(cfg.get("a_passes_per_step") or 0.0)
    agg = d["value"] * passes * m * n * 4  # bytes of A streamed per second by the two ranks together
    return {"value": d["value"], "unit": "it/s", "steps": d.get("steps"), "ms_per_step": d.get("ms_per_step"),
            "config": {"workload": "FFB LASSO m=%d n=%d Float32, fixed step, rows of A over 2 PROCESSES sharing this device%s" % (
                           m, n, " as a row team" if teams else " (two sweeps + all-reduce)"),
                       "m": m, "n": n, "m_per_rank": m // 2, "a_passes_per_step": passes, "row_teams": bool(cfg.get("row_teams")),
                       "sweep_fallbacks": cfg.get("sweep_fallbacks"), "row_team_selftest": cfg.get("row_team_selftest"),
                       "row_team_stats": cfg.get("row_team_stats"), "row_team_geometry": cfg.get("row_team_geometry"),
                       "row_team_geometries_tried": cfg.get("row_team_geometries_tried"), "final": cfg.get("final")},
            "roofline": {"bound": "hbm", "kernel": "gemv_tn (row team, 2 processes)" if teams else "gemv_n_partial + gemv_t (2 processes)",
                         "avg_launch_ms": (d.get("roofline") or {}).get("avg_launch_ms"), "achieved": round(agg / 1e9, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(agg / 1e9 / HBM_PEAK_GBS, 4),
                         "note": "aggregate over the two ranks: it/s x reads of A per iteration x m n s"}}


ROW_TEAM_RECORDS = (("rows_strong_teams", "strong"), ("config5_weak_rows_teams", "weak"))


def row_team_record_is_clean(r):
    """(ok, reason): may this row-team record stand as the top-level record?  It must have been measured, as a row team, with the
    attach-time scalar exchange right on EVERY rank, one read of the block per step and no sweep redone with two."""
    if not isinstance(r, dict) or "value" not in r:
        return False, "the row-team record was not measured: %s" % str((r or {}).get("error", "no record"))[:160]
    cfg = r.get("config") or {}
    if not cfg.get("row_teams"):
        return False, "the record did not run as a row team"
    if cfg.get("row_team_selftest_all_ranks") is not True:
        return False, "the attach-time self-test of the inboxes failed on some rank (rank 0: %s)" % cfg.get("row_team_selftest")
    if cfg.get("sweep_fallbacks"):
        return False, "%d step(s) of the row-team run fell back to two sweeps (a bounded wait expired)" % cfg["sweep_fallbacks"]
    passes = cfg.get("a_passes_per_step")
    if passes is None or passes > 1.05:
        return False, "the row-team run read its block %.2f times per step" % (passes or 0.0)
    return True, "self-test ok on every rank, %d steps at %.2f reads of the block per step, no fallback" % (r.get("steps", 0), passes)


def promote_row_team_record(args, job):
    """The contract layout's top-level record: the row-team record of the SAME problem and the same K steps when it is clean
    (row_team_record_is_clean), else the two-sweep record measured in this process group; `config.row_layout` /
    `config.row_layout_reason` say which and why, and after an upgrade the two-sweep record stays in the line as
    `rows_two_sweeps`."""
    key = "rows_strong_teams" if args.scaling == "strong" else "config5_weak_rows_teams"
    cand = job.extra.get(key)
    ok, why = row_team_record_is_clean(cand)
    if ok and cand.get("steps") != args.steps:
        ok, why = False, "the row-team record ran %s steps, not the line's %d" % (cand.get("steps"), args.steps)
    if not ok:
        job.layout_note = {"row_layout": "two_sweeps", "row_layout_reason": why}
        return False
    two = job.main_rec
    job.extra = dict([("rows_two_sweeps", two)] + [(k_, v) for k_, v in job.extra.items() if k_ != key])
    job.main_rec = cand
    job.layout_note = {"row_layout": "row_teams", "row_layout_reason": why,
                       "rows_two_sweeps_it_s": two.get("value")}
    return True


LATE_THRESHOLD = 0.05  # of the wave-steps of a row-team record: above it the child tries the fewest-transactions geometry as well


def late_fraction(stats, geom, n):
    """late wave-steps / wave-steps of the row-team sweeps counted in `stats` (pg_ctx_row_team_stats): every sweep visits each of the
    ceil(n / C) column groups once, with W waves"""
    if not stats or not geom or not stats.get("sweeps"):
        return None
    wave_steps = stats["sweeps"] * geom.get("W", 1) * -(-n // max(geom.get("C", 1), 1))
    return round(stats.get("late_waves", 0) / max(wave_steps, 1), 5)


def row_team_second_geometry(first, rerun, pa, ctx, D, world):
    """AT MOST two geometries for a row-team record, the second only when the first one's granules came late (more than LATE_THRESHOLD
    of its wave-steps): one post per two steps -- half the fabric transactions, pg_ctx_row_team_tune "PAIR" -- where the sweep has
    that form (the one-wave sweep, K1=1).  The ranks decide together (the largest late fraction of any rank; the faster record by
    rank 0's clock).  Returns the record to keep; `config.row_team_geometries_tried` says what both did."""
    import torch.distributed as dist

    late = (first["config"].get("row_team_stats") or {}).get("late_fraction") or 0.0
    late = D.reduce_scalar(float(late), dist.ReduceOp.MAX) if world > 1 else late
    threshold = float(os.environ.get("PG_BENCH_LATE_THRESHOLD", LATE_THRESHOLD))  # (tests force the second try with -1)
    geom = str(first["config"].get("row_team_geometry"))
    if not (late > threshold and "K1=1" in geom and "PAIR=1" not in geom):
        return first
    pa.row_team_tune(ctx, PAIR=1)
    try:
        second = rerun()
    finally:
        pa.row_team_tune(ctx, PAIR=2)  # (back to one post per step for whatever follows)
    late2 = (second["config"].get("row_team_stats") or {}).get("late_fraction") or 0.0
    late2 = D.reduce_scalar(float(late2), dist.ReduceOp.MAX) if world > 1 else late2
    faster = second["value"] > first["value"]
    if world > 1:
        faster = D.reduce_scalar(1.0 if faster else 0.0, dist.ReduceOp.MIN) > 0.5
    keep = second if faster else first
    keep["config"]["row_team_geometries_tried"] = ("one post per step: %.4g it/s, %.2f %% of the wave-steps late; one post per two steps: %.4g it/s, %.2f %% late"
                                                  % (first["value"], 100.0 * late, second["value"], 100.0 * late2))
    return keep


def row_team_child(args, job, wd, pa, ctx, D, world, rank, m_base, n, dtype):
    """`--row-teams-child`: the two row-team records (north_star's row layout at ONE read of A per iteration: the ranks push
    per-column partial dots into each other's IPC-mapped inbox inside the sweep, csrc/pg_gemv_tn4.hip) in a process group of
    their own.  They are the only part of an N > 1 line that has never run on more than one GPU; a GPU fault in here must
    not take the measured line with it, so every rank of the job starts THIS program as a child after its own records (never
    an exec), waits for it under a deadline and merges what rank 0's child printed."""
    import torch.distributed as dist

    records = {}
    sub_steps = max(4, min(args.steps, 20))
    for key, scaling in ROW_TEAM_RECORDS:
        # the record of the line's own top-level problem runs the line's K steps after W warm-up steps (it may become the
        # top-level record: promote_row_team_record), the other one the short form of every sub-record
        top = scaling == args.scaling
        wd.enter(key, (args.record_timeout if top else args.sub_record_timeout))
        t_sub = time.perf_counter()
        try:
            P2 = setup_lasso(pa, ctx, D, m_base * world if scaling == "weak" else m_base, n, dtype, args.seed, "rows", "fixed", row_teams=True)
            records[key] = run_ffb(pa, ctx, D, P2, "fixed", "one", args.steps if top else sub_steps, args.warmup if top else 3,
                                   args.kernel_events, scaling=scaling)
            if top:
                records[key] = row_team_second_geometry(records[key], lambda: run_ffb(pa, ctx, D, P2, "fixed", "one", args.steps, args.warmup, args.kernel_events,
                                                                                       scaling=scaling), pa, ctx, D, world)
            del P2
        except Exception as e:  # noqa: BLE001 -- reported in the record; the ranks may be out of step now, so the other one is skipped
            traceback.print_exc()
            records[key] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            records[key]["wall_s"] = round(time.perf_counter() - t_sub, 2)
            break
        records[key]["wall_s"] = round(time.perf_counter() - t_sub, 2)
        import gc

        gc.collect()
        ctx.sync()
        time.sleep(0.5)
    wd.enter("finalize", 60.0)
    if rank == 0:
        os.write(job.json_fd, (json.dumps({"row_team_records": records}) + "\n").encode())
    try:
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass
    wd.close()
    return 0


def row_team_records_in_a_child(args, job, wd, ctx, world, rank, freed_bytes=0):
    """parent side of row_team_child: free this rank's blocks, start the child (same ranks, another rendezvous port), wait,
    merge rank 0's records into the line.  Whatever the child does -- refuses, times out, dies of a GPU fault -- the records
    measured before it stand and the exit code is not its business."""
    import gc

    gc.collect()
    ctx.sync()
    if freed_bytes > (1 << 30):  # the block this rank just gave back is still being cleared in the background: kernels running
        time.sleep(min(6.0, freed_bytes / 30e9 + 0.3))  # meanwhile lose a few percent (see `settle` in run_rank) -- the child's would
    budget = args.record_timeout + args.sub_record_timeout + args.init_timeout + 60.0
    wd.enter("row_teams_child", budget + 30.0, stall=False)
    t0 = time.perf_counter()
    # the child ranks rendezvous on a store of their own: not the launcher's agent store (TORCHELASTIC_USE_AGENT_STORE would make
    # rank 0 a client of a server nobody runs on the new port)
    env = {k_: v for k_, v in os.environ.items() if not k_.startswith("TORCHELASTIC_")}
    # ... on a port rank 0 finds free NOW and tells the others (a fixed offset from the parent's port was taken once in a while:
    # a self-launched job's port is an ephemeral one and the parent's own gloo pairs live right beside it -- EADDRINUSE on rank 0
    # while its peers waited out their rendezvous)
    import torch.distributed as dist

    box = [None]
    if rank == 0:
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s_:
            s_.bind(("127.0.0.1", 0))
            box[0] = s_.getsockname()[1]
    try:
        dist.broadcast_object_list(box, src=0)
    except Exception:  # noqa: BLE001 -- fall back to the fixed offset
        box[0] = None
    env["MASTER_PORT"] = str(box[0] or int(os.environ.get("MASTER_PORT", "29577")) + 23)
    env.pop("PG_BENCH_ARGV", None)
    own = json.loads(os.environ["PG_BENCH_ARGV"]) if (len(sys.argv) == 1 and os.environ.get("PG_BENCH_ARGV")) else sys.argv[1:]
    argv = [a for a in own if a != "--row-teams"]
    cmd = [sys.executable, os.path.abspath(__file__)] + argv + ["--row-teams-child"]
    out, err, note = "", "", None
    try:
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, start_new_session=True, text=True)
        try:
            out, err = proc.communicate(timeout=budget)
            if proc.returncode != 0:
                note = "the row-team child exited with code %d" % proc.returncode
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            out, err = proc.communicate()
            note = "the row-team child did not finish within %.0f s" % budget
    except Exception as e:  # noqa: BLE001
        note = "the row-team child could not be started: %s" % str(e)[:200]
    if err:
        sys.stderr.write(err[-4000:])
    # the ranks meet again before the line is finalised (still under this stage's deadline): a child that failed at once on ONE
    # rank leaves its peers waiting out their rendezvous, and finalize's short deadline is not meant to cover that
    try:
        dist.barrier()
    except Exception:  # noqa: BLE001
        pass
    if rank != 0:
        return
    recs = None
    for ln in out.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and "row_team_records" in ln:
            try:
                recs = json.loads(ln)["row_team_records"]
            except ValueError:
                pass
    for key, _ in ROW_TEAM_RECORDS:
        if recs and key in recs:
            job.extra[key] = recs[key]
        else:
            job.extra[key] = {"error": note or "no record from the row-team child", "stderr_tail": [l_ for l_ in err.splitlines() if l_.strip()][-4:]}
    job.extra["row_teams_child_s"] = round(time.perf_counter() - t0, 2)


def cpu_leg(args, P, m_glob, n, es):
    # the whole matrix when the host can hold it (3x headroom), else a column sample scaled linearly in n
    need = 3 * m_glob * n * es
    lim = _host_memory_limit()
    if args.cpu_baseline == "full" or (args.cpu_baseline == "auto" and lim is not None and lim >= need):
        return cpu_baseline_full(P["A"], P["b"], P["lam"], P["Lf"])
    return cpu_baseline(m_glob, n, args.cpu_sample_cols, args.cpu_steps, args.seed)


if __name__ == "__main__":
    main()
