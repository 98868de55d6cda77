#!/usr/bin/env python3
"""Rate of the two-point sweep (pg_mat_fused_tn_pair) against the single sweep (pg_mat_fused_tn) at BASELINE config 4's size
(16384 x 10^6 Float32): event-pair kernel times of `gemv_tn` launches, per geometry of the pair kernel (PG_TUNE: PG_TNP2_W, PG_TNP2_C).
    python scripts/r5_pair_sweep_rate.py [--n 1000000] [--reps 12]"""
import argparse
import json
import os
import sys

import numpy as np

os.environ["PG_TUNE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=16384)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--variants", default="8:3")
    args = ap.parse_args()
    import proximalalgorithms.jl_amd as pa

    m, n, dtype = args.m, args.n, np.float32
    ctx = pa.get_context()
    A = pa.HIPMatrix.synthetic(m, n, dtype, seed=0)
    rng = np.random.default_rng(1)
    r = [pa.HIPVector.from_numpy(rng.standard_normal(m).astype(dtype)) for _ in range(3)]
    x = [pa.HIPVector.from_numpy(rng.standard_normal(n).astype(dtype)) for _ in range(3)]
    g = pa.NormL1(dtype(0.05))
    o1 = [x[0].similar() for _ in range(4)] + [r[0].similar()]
    o2 = [x[0].similar() for _ in range(4)] + [r[0].similar()]
    o3 = [x[0].similar() for _ in range(4)] + [r[0].similar()]
    bytes_a = m * n * 4

    def timed(fn):
        for _ in range(3):
            fn()
        ctx.profile(True, kernels=("gemv_tn",))
        ctx.profile_reset()
        for _ in range(args.reps):
            fn()
        ctx.sync()
        cnt, ms = ctx.profile_read()["gemv_tn"]
        ctx.profile(False)
        return ms / cnt

    t1 = timed(lambda: A.fused_tn(r[0], x[0], 0.3, g, *o1))
    print(json.dumps({"kernel": "single sweep (gemv_tnm<16,2,4,2>)", "ms": round(t1, 4), "TBps": round(bytes_a / t1 / 1e9, 3), "of_8TBps": round(bytes_a / t1 / 8e9, 4)}), flush=True)
    for v in args.variants.split(","):
        w, c = v.split(":")
        os.environ["PG_TNP2_W"], os.environ["PG_TNP2_C"] = w, c
        try:
            t2 = timed(lambda: A.fused_tn_pair(r[0], x[0], r[1], x[1], 0.3, g, o1, o2))
        except pa.ProxGradError as e:
            print(json.dumps({"kernel": "pair W=%s C=%s" % (w, c), "error": str(e)[:200]}), flush=True)
            continue
        print(json.dumps({"kernel": "pair sweep W=%s C-code=%s" % (w, c), "ms": round(t2, 4), "TBps": round(bytes_a / t2 / 1e9, 3),
                          "of_8TBps": round(bytes_a / t2 / 8e9, 4), "cost_in_single_sweeps": round(t2 / t1, 3)}), flush=True)
    t3 = timed(lambda: A.fused_tn_trio(r, x, 0.3, g, (o1, o2, o3)))
    print(json.dumps({"kernel": "three-point sweep (gemv_tnm_trio<8,2,8,3>)", "ms": round(t3, 4), "TBps": round(bytes_a / t3 / 1e9, 3),
                      "of_8TBps": round(bytes_a / t3 / 8e9, 4), "cost_in_single_sweeps": round(t3 / t1, 3)}), flush=True)


if __name__ == "__main__":
    main()
