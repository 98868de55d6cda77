#!/usr/bin/env python3
"""Round 3: exact-U instantiations of the one-workgroup sweep for mid-length columns (csrc/pg_gemv_tn3.hip) against the
default dispatch, interleaved A/B (median of five rounds), plus a correctness check of each candidate against float64 numpy.
    python scripts/r3_mid_sweep.py check
    python scripts/r3_mid_sweep.py ab [m n]..."""
import os

os.environ.setdefault("PG_TUNE", "1")  # the library reads its tuning variables only when this is set

import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import proximalalgorithms.jl_amd as pa  # noqa: E402
import r2_tn_check as r2  # noqa: E402

r2.KNOBS = r2.KNOBS + ("PG_TN_U", "PG_TN_DB", "PG_TN_MIDK")


def mid(U, C, W, db, bpc=1, k=1):
    return dict(PG_TN_KERNEL="mid", PG_TN_U=str(U), PG_TN_C=str(C), PG_TN_WAVES=str(W), PG_TN_DB=str(db), PG_TN_BLOCKS_PER_CU=str(bpc),
                PG_TN_MIDK=str(k))


def candidates(nrg, both=True):
    """(label, env): the default dispatch, then gemv_tnm (k = 1: branch-free tile loads, scalar x_j / z_old_j loads) and the
    exact-U instantiations of gemv_tn_kernel (k = 0) in the geometries that fit nrg row groups"""
    c = [("default", {}), ("gemv_tn (round 2)", dict(PG_TN_KERNEL="wg"))]
    geo = []
    if 17 <= nrg <= 32:
        U = (nrg + 3) // 4
        geo += [(U, 4, 4, db, 1) for db in (0, 1)]
        U8 = (nrg + 7) // 8
        geo += [(U8, 8, 8, 0, 1), (U8, 4, 8, 0, 1), (U8, 4, 8, 1, 1)]
    if 33 <= nrg <= 64:
        U = (nrg + 3) // 4
        geo += [(U, 2, 4, db, 1) for db in (0, 1)]
        if U in (10, 12):
            geo += [(U, 4, 4, 0, 1)]
    if 65 <= nrg <= 128:
        U = (nrg + 7) // 8
        geo += [(U, C, 8, 0, 1) for C in (1, 2) if not (C == 2 and U > 13)]
    for (U, C, W, db, bpc) in geo:
        c.append((f"tnm U={U} C={C} W={W} tiles={db + 1} wg/CU={bpc}", mid(U, C, W, db, bpc, 1)))
    return c


def cmd_ab(shapes):
    ctx = pa.get_context()
    g = pa.NormL1(0.3)
    for (m, n) in shapes:
        A, f, x, vs = r2.setup(m, n)
        nbytes = m * n * 4
        nrg = (m * 4 + 1023) // 1024
        cands = candidates(nrg)
        got = {k: [] for k, _ in cands}
        for _ in range(5):
            for k, env in cands:
                r2.clear()
                os.environ.update(env)
                try:
                    got[k].append(nbytes / (r2.time_pass(f, x, vs, g, ctx) * 1e-3) / 1e9)
                except pa.ProxGradError:
                    pass
        r2.clear()
        print(f"=== {m}x{n} f32 ({nrg} row groups) ===")
        for k, v in sorted(got.items(), key=lambda kv: -np.median(kv[1]) if kv[1] else 0):
            if v:
                print(f"   {k:42s} {np.median(v):6.0f} GB/s  ({np.median(v) / 8000:.3f})")
        del f, A


def cmd_check():
    ok = True
    for dtype in (np.float32, np.float64):
        rpg = 1024 // np.dtype(dtype).itemsize
        for nrg in (17, 20, 24, 29, 32, 33, 37, 40, 44, 47, 52, 57, 60, 64, 65, 72, 81, 90, 100, 104, 113, 120, 128):
            for m in (nrg * rpg, nrg * rpg - 3):
                for name, env in candidates(nrg, both=False)[1:]:
                    if "wg/CU=2" in name:
                        continue
                    for n in (37, 1000):
                        try:
                            ok &= r2.check_one(m, n, dtype, env)
                        except pa.ProxGradError as e:
                            print("  skip", env, str(e)[:80])
                            r2.clear()
    print("ALL OK" if ok else "FAILURES")
    return 0 if ok else 1


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "ab"
    rest = [int(v) for v in sys.argv[2:]]
    shapes = [(rest[i], rest[i + 1]) for i in range(0, len(rest) - 1, 2)]
    if cmd == "check":
        sys.exit(cmd_check())
    total = 8192 * 262144  # elements: 8 GiB of Float32 per shape
    cmd_ab(shapes or [(256 * g, total // (256 * g)) for g in (17, 20, 24, 28, 32, 33, 36, 40, 44, 48, 52, 56, 60, 64, 65, 72, 80, 88, 96,
                                                                104, 112, 120, 128)])
