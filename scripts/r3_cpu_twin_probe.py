#!/usr/bin/env python3
"""Probe of the host side of the GPU box: how the C / OpenMP CPU twin scales with threads there (bench.py's CPU leg)."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), {k: v for k, v in os.environ.items() if "OMP" in k or "KMP" in k})
try:
    print(open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("cpu.max:", e)
print(subprocess.run("lscpu | egrep 'Model name|Socket|NUMA|Thread|Core' ; numactl -H 2>/dev/null | head -8", shell=True, capture_output=True, text=True).stdout)
with_torch = len(sys.argv) > 1 and sys.argv[1] == "torch"
if with_torch:
    import torch
    print("torch threads", torch.get_num_threads())
from oracle import cpu_twin
m, n = 16384, 131072  # 8 GiB
t0 = time.time()
A = np.empty((m, n), np.float32, order="F")
A[:] = 0.001
b = np.ones(m, np.float32)
print("alloc+fill %.1f s" % (time.time() - t0))
for thr in (None, 128, 64, 32, 16, 8, 4, 1):
    z, fx, sec, t = cpu_twin.ffb(A, b, 0.1, 4.0 * n / m, 2, threads=thr)
    print("threads asked", thr, "max", t, "GB/s %.1f" % (2 * m * n * 4 * 2 / sec / 1e9), flush=True)
