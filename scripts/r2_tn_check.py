#!/usr/bin/env python3
"""Round-2 bring-up of the two new single-sweep geometries (csrc/pg_gemv_tn2.hip):
    python scripts/r2_tn_check.py check          correctness of wave / wg / team kernels against float64 numpy
    python scripts/r2_tn_check.py short          sweep of the short-column (one wave per column group) kernel
    python scripts/r2_tn_check.py team [m n]...  sweep of the long-column (workgroup team) kernel
Prints plain text; the tables kept under profiles/ come from here."""
import os

os.environ.setdefault("PG_TUNE", "1")  # the library reads its tuning variables only when this is set

import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proximalalgorithms.jl_amd as pa  # noqa: E402

KNOBS = ("PG_TN_KERNEL", "PG_TNW_C", "PG_TNW_WPB", "PG_TNW_DB", "PG_TNW_WAVES_PER_CU", "PG_TNT_C", "PG_TNT_LAG", "PG_TNT_PF", "PG_TNT_WAVES", "PG_TNT_U", "PG_TN_TEAM",
         "PG_TN_TEAMS", "PG_TN_WAVES", "PG_TN_C", "PG_TN_BLOCKS_PER_CU", "PG_TNC_WAVES", "PG_TNC_C", "PG_TNC_DB",
         "PG_TNC_BLOCKS_PER_CU")


def clear():
    for k in KNOBS:
        os.environ.pop(k, None)


def reference(A, b, x, zold, gamma, beta, lam_l1, dtype):
    A64, x64 = A.astype(np.float64), x.astype(np.float64)
    r = A64 @ x64 - b
    g = A64.T @ r
    y = x64 - gamma * g
    t = gamma * lam_l1
    z = np.sign(y) * np.maximum(np.abs(y) - t, 0.0)
    res = x64 - z
    v = z + beta * (z - zold.astype(np.float64))
    rn = A64 @ v - b
    return g, y, z, res, v, 0.5 * rn @ rn, lam_l1 * np.abs(z).sum(), np.abs(res).max(), g @ res, res @ res


def check_one(m, n, dtype, env, seed=0):
    clear()
    os.environ.update(env)
    rng = np.random.default_rng(seed)
    A = np.asfortranarray(rng.standard_normal((m, n)).astype(dtype) / np.sqrt(m).astype(dtype))
    b = rng.standard_normal(m).astype(dtype)
    x = (0.3 * rng.standard_normal(n)).astype(dtype)
    zold = (0.3 * rng.standard_normal(n)).astype(dtype)
    gamma, beta, lam = 0.37, 0.61, 0.05
    f = pa.LeastSquares(A, b)
    xd, zd = pa.HIPVector.from_numpy(x), pa.HIPVector.from_numpy(zold)
    f(xd)  # residual of x
    outs = [xd.similar() for _ in range(5)]
    sc = f.fused_pass(xd, zd, gamma, beta, pa.NormL1(lam), *outs)
    ref = reference(A, b.astype(np.float64), x, zold, dtype(gamma), dtype(beta), float(dtype(lam)), dtype)
    tol = 3e-5 if dtype == np.float32 else 1e-12
    worst = 0.0
    for name, got, want in zip(("g", "y", "z", "res", "v"), outs, ref[:5]):
        err = np.max(np.abs(got.numpy().astype(np.float64) - want)) / max(1.0, np.max(np.abs(want)))
        worst = max(worst, err)
    rnext = f.residual().numpy().astype(np.float64)
    err_r = np.max(np.abs(rnext - (A.astype(np.float64) @ ref[4] - b))) / max(1.0, np.max(np.abs(rnext)))
    worst = max(worst, err_r)
    scal = [abs(float(s) - w) / max(1.0, abs(w)) for s, w in zip(sc, ref[5:])]
    worst = max(worst, max(scal))
    ok = worst <= tol * 10
    print(f"  {'ok  ' if ok else 'FAIL'} m={m:7d} n={n:6d} {np.dtype(dtype).name} {env}  worst rel err {worst:.2e}")
    clear()
    return ok


def cmd_check():
    ok = True
    for dtype in (np.float32, np.float64):
        rpg = 1024 // np.dtype(dtype).itemsize
        for m in (1, 5, rpg - 1, rpg, 2 * rpg + 3, 4 * rpg, 7 * rpg + 1, 8 * rpg):
            for n in (1, 3, 37, 1000):
                U = 1
                while U * rpg < m:
                    U *= 2
                for env in ({"PG_TN_KERNEL": "wave"}, {"PG_TN_KERNEL": "wave", "PG_TNW_WPB": "1"},
                            {"PG_TN_KERNEL": "wave", "PG_TNW_DB": "0"},
                            {"PG_TN_KERNEL": "wave", "PG_TNW_C": str(min(32 // U, 16 if dtype == np.float64 else 32))}):
                    ok &= check_one(m, n, dtype, env)
        # teams: forced on short columns (partly idle members), then real long columns
        for m, n, env in ((3 * rpg, 50, {"PG_TN_KERNEL": "team", "PG_TN_TEAM": "3"}),
                          (100 * rpg, 77, {"PG_TN_KERNEL": "team"}),
                          (130 * rpg + 5, 40, {}),
                          (256 * rpg, 40, {}), (256 * rpg, 333, {"PG_TNT_PF": "1"}),
                          (256 * rpg, 333, {"PG_TNT_LAG": "1"}), (256 * rpg, 333, {"PG_TNT_LAG": "0"}),
                          (256 * rpg, 333, {"PG_TNT_WAVES": "8", "PG_TNT_U": "8"}),
                          (256 * rpg, 333, {"PG_TNT_WAVES": "8", "PG_TNT_U": "8", "PG_TNT_PF": "1"}),
                          (256 * rpg, 333, {"PG_TNT_WAVES": "8", "PG_TNT_U": "4"}),
                          (300 * rpg + 17, 700, {}), (512 * rpg, 24, {}), (1024 * rpg, 9, {})):
            ok &= check_one(m, n, dtype, env)
        # medium columns: waves share the column group, lane-parallel epilogue (gemv_tnc)
        for m in (3 * rpg, 9 * rpg + 1, 16 * rpg, 17 * rpg, 32 * rpg):
            for n in (1, 37, 1000):
                nrg = (m + rpg - 1) // rpg
                for env in ({"PG_TN_KERNEL": "coop"}, {"PG_TN_KERNEL": "coop", "PG_TNC_DB": "1", "PG_TNC_C": str(16 // max(1, (nrg + 7) // 8 if nrg > 16 else (nrg + 3) // 4) or 2)}):
                    try:
                        ok &= check_one(m, n, dtype, env)
                    except pa.ProxGradError as e:
                        print("  skip", env, str(e)[:80])
                        clear()
        # the one-workgroup kernel still answers (regression of the split into translation units)
        for m, n in ((16 * rpg, 100), (64 * rpg, 30), (128 * rpg, 20)):
            ok &= check_one(m, n, dtype, {"PG_TN_KERNEL": "wg"})
    print("ALL OK" if ok else "FAILURES")
    return 0 if ok else 1


def time_pass(f, x, vs, g, ctx, reps=10):
    for _ in range(2):
        f.fused_pass(x, x, 0.01, 0.5, g, *vs)
    ctx.profile(True, kernels=("gemv_tn",))
    ctx.profile_reset()
    for _ in range(reps):
        f.fused_pass(x, x, 0.01, 0.5, g, *vs)
    cnt, ms = ctx.profile_read()["gemv_tn"]
    ctx.profile(False)
    return ms / cnt


def setup(m, n, dtype=np.float32):
    A = pa.HIPMatrix.synthetic(m, n, dtype, seed=0)
    b = pa.HIPVector.from_numpy(np.random.default_rng(1).standard_normal(m).astype(dtype))
    f = pa.LeastSquares(A, b)
    x = pa.HIPVector.from_numpy((0.01 * np.random.default_rng(2).standard_normal(n)).astype(dtype))
    f(x)
    vs = [x.similar() for _ in range(5)]
    return A, f, x, vs


def cmd_short(shapes):
    ctx = pa.get_context()
    g = pa.NormL1(0.3)
    for (m, n) in shapes:
        A, f, x, vs = setup(m, n)
        nbytes = m * n * 4
        nrg = (m * 4 + 1023) // 1024
        U = 1
        while U < nrg:
            U *= 2
        clear()
        os.environ["PG_TN_KERNEL"] = "wg"
        base = nbytes / (time_pass(f, x, vs, g, ctx) * 1e-3) / 1e9
        res = []
        for C in (2, 4, 8, 16, 32):
            for wpb in (1, 2, 4, 8):
                for db in (0, 1):
                    for wpc in (2, 3, 4, 5, 6, 8, 12, 16):
                        if wpc < wpb or (wpc % wpb and wpb > 1):
                            continue
                        clear()
                        os.environ.update(PG_TN_KERNEL="wave", PG_TNW_C=str(C), PG_TNW_WPB=str(wpb), PG_TNW_DB=str(db),
                                          PG_TNW_WAVES_PER_CU=str(wpc))
                        try:
                            res.append((nbytes / (time_pass(f, x, vs, g, ctx) * 1e-3) / 1e9, C, wpb, db, wpc))
                        except pa.ProxGradError:
                            break  # not instantiated
        clear()
        dflt = nbytes / (time_pass(f, x, vs, g, ctx) * 1e-3) / 1e9
        res.sort(reverse=True)
        print(f"=== {m}x{n} f32 (U={U}) ===  one-workgroup kernel {base:6.0f} GB/s ; default dispatch {dflt:6.0f} GB/s")
        for gb, C, wpb, db, wpc in res[:8]:
            print(f"   wave kernel C={C:2d} waves/wg={wpb} double_buffer={db} waves/CU={wpc:2d}: {gb:6.0f} GB/s")
        del f, A


def cmd_mid(shapes):
    ctx = pa.get_context()
    g = pa.NormL1(0.3)
    for (m, n) in shapes:
        A, f, x, vs = setup(m, n)
        nbytes = m * n * 4
        nrg = (m * 4 + 1023) // 1024
        clear()
        os.environ["PG_TN_KERNEL"] = "wg"
        base = nbytes / (time_pass(f, x, vs, g, ctx) * 1e-3) / 1e9
        res = []
        for W in (2, 4, 8):
            U = 1
            while U * W < nrg:
                U *= 2
            for C in (2, 4, 8, 16):
                for db in (0, 1):
                    for bpc in (1, 2, 3, 4):
                        clear()
                        os.environ.update(PG_TN_KERNEL="coop", PG_TNC_WAVES=str(W), PG_TNC_C=str(C), PG_TNC_DB=str(db),
                                          PG_TNC_BLOCKS_PER_CU=str(bpc))
                        try:
                            res.append((nbytes / (time_pass(f, x, vs, g, ctx) * 1e-3) / 1e9, W, U, C, db, bpc))
                        except pa.ProxGradError:
                            break
        clear()
        dflt = nbytes / (time_pass(f, x, vs, g, ctx) * 1e-3) / 1e9
        res.sort(reverse=True)
        print(f"=== {m}x{n} f32 ({nrg} row groups) ===  gemv_tn (round 1 geometry) {base:6.0f} GB/s ; default dispatch {dflt:6.0f} GB/s")
        for gb, W, U, C, db, bpc in res[:8]:
            print(f"   gemv_tnc waves={W} U={U} C={C:2d} double_buffer={db} workgroups/CU={bpc}: {gb:6.0f} GB/s")
        del f, A


def cmd_ab(shapes):
    """Interleaved A/B of the candidate geometries for 9..32 row groups (median of five rounds each)."""
    ctx = pa.get_context()
    g = pa.NormL1(0.3)
    for (m, n) in shapes:
        A, f, x, vs = setup(m, n)
        nbytes = m * n * 4
        nrg = (m * 4 + 1023) // 1024
        coop = lambda C, db: dict(PG_TN_KERNEL="coop", PG_TNC_WAVES="8", PG_TNC_C=str(C), PG_TNC_DB=str(db))
        cands = [("default", {}), ("wg", dict(PG_TN_KERNEL="wg"))]
        if 9 <= nrg <= 32:
            cands += [(f"coop C={C} db={db}", coop(C, db)) for C in ((8, 16) if nrg <= 16 else (4, 8)) for db in (0, 1)]
        if 17 <= nrg <= 32:
            cands += [(f"wg W=4 C={C}", dict(PG_TN_KERNEL="wg", PG_TN_WAVES="4", PG_TN_C=str(C))) for C in (2, 4)]
        if nrg > 32:
            cands += [("team", dict(PG_TN_KERNEL="team"))]
        if nrg > 64 and nrg % 64 == 0:
            cands += [("team PF=1", dict(PG_TN_KERNEL="team", PG_TNT_PF="1"))]
        got = {k: [] for k, _ in cands}
        for _ in range(5):
            for k, env in cands:
                clear()
                os.environ.update(env)
                try:
                    got[k].append(nbytes / (time_pass(f, x, vs, g, ctx) * 1e-3) / 1e9)
                except pa.ProxGradError:
                    pass
        print(f"=== {m}x{n} f32 ({nrg} row groups) === " + " ; ".join(f"{k}: {np.median(v):5.0f}" for k, v in got.items() if v))
        del f, A


W8 = {"PG_TNT_WAVES": "8"}
TEAM_ENVS = ({}, {"PG_TNT_PF": "1"}, {"PG_TNT_LAG": "1"}, {"PG_TNT_LAG": "0"}, dict(W8, PG_TNT_U="8"), dict(W8, PG_TNT_U="8", PG_TNT_PF="1"),
             dict(W8, PG_TNT_U="4"), {})


def cmd_team(shapes, envs=TEAM_ENVS):
    ctx = pa.get_context()
    g = pa.NormL1(0.3)
    for (m, n) in shapes:
        A, f, x, vs = setup(m, n)
        nbytes = m * n * 4
        print(f"=== {m}x{n} f32 ({nbytes / 2**30:.1f} GiB) ===")
        for env in envs:
            clear()
            if m * 4 <= 128 * 1024 and "PG_TN_TEAM" not in env:
                env = dict(env, PG_TN_KERNEL="team")
            os.environ.update(env)
            try:
                ms = time_pass(f, x, vs, g, ctx, reps=6)
                print(f"   {str(env):60s} {ms:8.3f} ms  {nbytes / (ms * 1e-3) / 1e9:6.0f} GB/s")
            except pa.ProxGradError as e:
                print(f"   {str(env):60s} error: {e}")
        clear()
        # the alternative: two sweeps (pass N + pass T)
        t0 = time.perf_counter()
        for _ in range(3):
            f.value_and_gradient(x, out=vs[0])
        ctx.sync()
        dt = (time.perf_counter() - t0) / 3
        print(f"   two separate sweeps (value_and_gradient): {dt * 1e3:8.3f} ms per evaluation = {2 * nbytes / dt / 1e9:6.0f} GB/s")
        del f, A


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "check"
    rest = [int(v) for v in sys.argv[2:]]
    shapes = [(rest[i], rest[i + 1]) for i in range(0, len(rest) - 1, 2)]
    if cmd == "check":
        sys.exit(cmd_check())
    if cmd == "short":
        cmd_short(shapes or [(512, 1 << 22), (1024, 1 << 21), (2048, 1 << 20)])
    if cmd == "ab":
        cmd_ab(shapes or [(256 * k, (1 << 23) // k // 8 * 8) for k in (9, 10, 12, 14, 16, 17, 20, 24, 28, 32)] + [(8192, 1 << 18)])
    if cmd == "mid":
        cmd_mid(shapes or [(4096, 1 << 19), (8192, 1 << 18), (3072, 1 << 19), (6144, 1 << 18)])
    if cmd == "team":
        cmd_team(shapes or [(131072, 131072), (65536, 262144)])
