#!/usr/bin/env python3
"""Parameter sweep of the two GEMV kernels (environment tunables of csrc/pg_gemv.hip) on an MI355X.
Usage: python scripts/tune_gemv.py [m n]...   -> prints GB/s per configuration and checks results agree."""
import itertools
import os

os.environ.setdefault("PG_TUNE", "1")  # the library reads its tuning variables only when this is set

import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import proximalalgorithms.jl_amd as pa  # noqa: E402


def timed(ctx, fn, reps, kname):
    fn()
    ctx.profile(True)
    ctx.profile_reset()
    for _ in range(reps):
        fn()
    cnt, ms = ctx.profile_read()[kname]
    ctx.profile(False)
    return ms / max(cnt, 1)


def setenv(**kw):
    for k, v in kw.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)


NRU = [(2, 2), (4, 1), (4, 2), (8, 1), (2, 4), (8, 2), (4, 4)]
NTW = [1]
NWPC = [3, 4, 5, 6, 7, 8, 10]
TCUW = [(2, 4, 8), (2, 4, 4), (2, 8, 4), (2, 16, 4), (2, 8, 2), (2, 16, 2), (1, 16, 2), (1, 16, 4), (2, 8, 1), (2, 16, 1),
        (4, 8, 2), (4, 4, 2), (4, 8, 4), (4, 4, 4), (2, 8, 8)]
TB = [1, 2]
REPS = 20


def main():
    shapes = [(16384, 1 << 20), (2048, 1 << 20), (8192, 262144)]
    if len(sys.argv) >= 3:
        shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
    ctx = pa.get_context()
    dt = np.float64 if os.environ.get("TUNE_DTYPE") == "f64" else np.float32
    for m, n in shapes:
        A = pa.HIPMatrix.synthetic(m, n, dt, seed=0)
        x = pa.HIPVector.from_numpy(np.random.default_rng(0).standard_normal(n).astype(dt))
        r = pa.HIPVector.from_numpy(np.random.default_rng(1).standard_normal(m).astype(dt))
        y, g = pa.HIPVector.empty(m, dt), pa.HIPVector.empty(n, dt)
        bytes_a = m * n * np.dtype(dt).itemsize
        print(f"=== m={m} n={n} ({bytes_a / 2**30:.1f} GiB) ===", flush=True)
        setenv(PG_N_R=None, PG_N_U=None, PG_N_TW=None, PG_N_WAVES_PER_CU=None)
        y_ref = A.mul(x, y).numpy().copy()
        res = []
        for (R, U), TW, W in itertools.product(NRU, NTW, NWPC):
            if m // 256 < R:
                continue
            setenv(PG_N_R=R, PG_N_U=U, PG_N_TW=TW, PG_N_WAVES_PER_CU=W)
            try:
                t = timed(ctx, lambda: A.mul(x, y), REPS, "gemv_n_partial")
            except Exception as e:
                print("  N", R, U, TW, W, "FAILED", e)
                continue
            err = float(np.max(np.abs(y.numpy() - y_ref)) / np.max(np.abs(y_ref)))
            res.append((bytes_a / t / 1e6, R, U, TW, W, err))
        res.sort(reverse=True)
        for gb, R, U, TW, W, err in res[:12]:
            print(f"  gemv_n R={R} U={U} TW={TW} waves/CU={W}: {gb:8.1f} GB/s  relerr={err:.1e}")
        print(f"  gemv_n worst: {res[-1]}")
        setenv(PG_N_R=None, PG_N_U=None, PG_N_TW=None, PG_N_WAVES_PER_CU=None)
        t = timed(ctx, lambda: A.mul(x, y), REPS, "gemv_n_partial")
        tf = ctx.profile_read()
        print(f"  gemv_n default: {bytes_a / t / 1e6:8.1f} GB/s")
        setenv(PG_T_C=None, PG_T_UR=None, PG_T_WAVES=None, PG_T_BLOCKS_PER_CU=None)
        g_ref = A.mul_adjoint(r, g).numpy().copy()
        res = []
        for (C, UR, W), B in itertools.product(TCUW, TB):
            lds = max(m * np.dtype(dt).itemsize // 1024, 1) * 1024
            if B * lds > 160 * 1024 or B * W > 32:
                continue
            setenv(PG_T_C=C, PG_T_UR=UR, PG_T_WAVES=W, PG_T_BLOCKS_PER_CU=B)
            try:
                t = timed(ctx, lambda: A.mul_adjoint(r, g), REPS, "gemv_t")
            except Exception as e:
                print("  T", C, UR, W, B, "FAILED", e)
                continue
            err = float(np.max(np.abs(g.numpy() - g_ref)) / np.max(np.abs(g_ref)))
            res.append((bytes_a / t / 1e6, C, UR, W, B, err))
        res.sort(reverse=True)
        for gb, C, UR, W, B, err in res[:12]:
            print(f"  gemv_t C={C} UR={UR} waves={W} blocks/CU={B}: {gb:8.1f} GB/s  relerr={err:.1e}")
        print(f"  gemv_t worst: {res[-1]}")
        setenv(PG_T_C=None, PG_T_UR=None, PG_T_WAVES=None, PG_T_BLOCKS_PER_CU=None)
        t = timed(ctx, lambda: A.mul_adjoint(r, g), REPS, "gemv_t")
        print(f"  gemv_t default: {bytes_a / t / 1e6:8.1f} GB/s")
        del A


if __name__ == "__main__":
    main()
