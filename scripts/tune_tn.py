#!/usr/bin/env python3
"""Sweep of the single-sweep kernel's launch geometry (environment tunables of csrc/pg_gemv.hip: PG_TN_C, PG_TN_BLOCKS):
python scripts/tune_tn.py [m n]...  -> GB/s of gemv_tn per configuration."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proximalalgorithms.jl_amd as pa

def main():
    shapes = [(16384, 1 << 20), (16384, 131072), (8192, 262144)]
    if len(sys.argv) >= 3:
        shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
    ctx = pa.get_context()
    for (m, n) in shapes:
        A = pa.HIPMatrix.synthetic(m, n, np.float32, seed=0)
        b = pa.HIPVector.from_numpy(np.random.default_rng(1).standard_normal(m).astype(np.float32))
        f = pa.LeastSquares(A, b)
        x = pa.HIPVector.from_numpy((0.01 * np.random.default_rng(2).standard_normal(n)).astype(np.float32))
        f(x)
        vs = [x.similar() for _ in range(5)]
        g = pa.NormL1(0.3)
        res = []
        nrg = m * 4 // 1024
        U = 1
        while U * 4 < nrg: U *= 2
        Cs = sorted({max(1, 16 // U), max(1, 32 // U)})
        for C in Cs:
            for blocks in (192, 256, 320, 384, 448, 512, 576, 640, 768, 1024):
                os.environ["PG_TN_C"] = str(C); os.environ["PG_TN_BLOCKS"] = str(blocks)
                try:
                    for _ in range(2): f.fused_pass(x, x, 0.01, 0.5, g, *vs)
                    ctx.profile(True, kernels=("gemv_tn",)); ctx.profile_reset()
                    for _ in range(8): f.fused_pass(x, x, 0.01, 0.5, g, *vs)
                    cnt, ms = ctx.profile_read()["gemv_tn"]; ctx.profile(False)
                    res.append((m * n * 4 / (ms / cnt * 1e-3) / 1e9, C, blocks))
                except Exception as e:
                    print("  failed", C, blocks, str(e)[:80])
        res.sort(reverse=True)
        print(f"=== {m}x{n} ===")
        for gb, C, blocks in res[:8]: print(f"  C={C} blocks={blocks}: {gb:7.0f} GB/s")
        print("  worst:", res[-1])
        os.environ.pop("PG_TN_C"); os.environ.pop("PG_TN_BLOCKS")
        del f, A

if __name__ == "__main__":
    main()
