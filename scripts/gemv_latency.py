#!/usr/bin/env python3
"""Latency of the two GEMV orientations (pg_mat_mul / pg_mat_mul_adjoint) on small and mid-size matrices: microseconds
per call from back-to-back launches (no host sync inside the timed loop).  python scripts/gemv_latency.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proximalalgorithms.jl_amd as pa

def main():
    ctx = pa.get_context()
    for dtype in (np.float64, np.float32):
        for (m, n) in ((5, 10), (50, 100), (500, 1000), (1000, 1000), (2000, 4000), (4000, 8000), (500, 500), (8192, 2048), (200, 20000)):
            A = pa.HIPMatrix.synthetic(m, n, dtype, seed=1)
            x = pa.HIPVector.zeros(n, dtype).fill_(0.5)
            r = pa.HIPVector.zeros(m, dtype).fill_(0.25)
            y, g = pa.HIPVector.empty(m, dtype), pa.HIPVector.empty(n, dtype)
            out = []
            for fn, a, o in ((A.mul, x, y), (A.mul_adjoint, r, g)):
                for _ in range(20): fn(a, o)
                ctx.sync(); t0 = time.perf_counter()
                K = 300
                for _ in range(K): fn(a, o)
                ctx.sync(); out.append((time.perf_counter() - t0) / K * 1e6)
            nbytes = m * n * np.dtype(dtype).itemsize
            print(f"{np.dtype(dtype).name} {m}x{n} ({nbytes/1e6:.2f} MB): mul {out[0]:.1f} us ({nbytes/out[0]/1e3:.0f} GB/s)  mul_adjoint {out[1]:.1f} us ({nbytes/out[1]/1e3:.0f} GB/s)")

if __name__ == "__main__":
    main()
