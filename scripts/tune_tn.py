#!/usr/bin/env python3
"""Sweep of the single-sweep kernel's launch geometry (environment tunables of csrc/pg_gemv.hip: PG_TN_WAVES, PG_TN_C,
PG_TN_BLOCKS_PER_CU): python scripts/tune_tn.py [m n]...  -> GB/s of gemv_tn per configuration, best first."""
import os, sys, numpy as np

os.environ.setdefault("PG_TUNE", "1")  # the library reads its tuning variables only when this is set

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proximalalgorithms.jl_amd as pa

KNOBS = ("PG_TN_WAVES", "PG_TN_C", "PG_TN_BLOCKS_PER_CU")

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
        for W in (1, 2, 4, 8):
            U = 1
            while U * W < nrg: U *= 2
            for C in sorted({max(1, 8 // U), max(1, 16 // U), max(1, 32 // U)}):
                for bpc in ((1, 2, 3, 4) if W >= 4 else (2, 4, 8, 16)):
                    os.environ.update(PG_TN_WAVES=str(W), PG_TN_C=str(C), PG_TN_BLOCKS_PER_CU=str(bpc))
                    try:
                        for _ in range(2): f.fused_pass(x, x, 0.01, 0.5, g, *vs)
                        ctx.profile(True, kernels=("gemv_tn",)); ctx.profile_reset()
                        for _ in range(10): f.fused_pass(x, x, 0.01, 0.5, g, *vs)
                        cnt, ms = ctx.profile_read()["gemv_tn"]; ctx.profile(False)
                        res.append((m * n * 4 / (ms / cnt * 1e-3) / 1e9, W, U, C, bpc))
                    except Exception as e:
                        ctx.profile(False)
                        if bpc == (1 if W >= 4 else 2): print(f"  W={W} U={U} C={C}: not instantiated")
                        break
        for k in KNOBS: os.environ.pop(k, None)
        for _ in range(2): f.fused_pass(x, x, 0.01, 0.5, g, *vs)
        ctx.profile(True, kernels=("gemv_tn",)); ctx.profile_reset()
        for _ in range(10): f.fused_pass(x, x, 0.01, 0.5, g, *vs)
        cnt, ms = ctx.profile_read()["gemv_tn"]; ctx.profile(False)
        res.sort(reverse=True)
        print(f"=== {m}x{n} ===   default geometry: {m * n * 4 / (ms / cnt * 1e-3) / 1e9:7.0f} GB/s")
        for gb, W, U, C, bpc in res[:10]: print(f"  W={W} U={U} C={C} blocks/CU={bpc}: {gb:7.0f} GB/s")
        print("  worst:", res[-1])
        del f, A

if __name__ == "__main__":
    main()
