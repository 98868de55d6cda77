#!/usr/bin/env python3
"""Parity of row-team sweep geometries chosen through the tuning variables (round 5: lag tiles in registers, latency injector; round 6: the one-wave sweep gemv_tnp1_kernel):
runs tests/tools/row_team.py (iterates of every rank against the CPU restatement on the whole matrix) once per case and prints
one summary line each.   python scripts/peer_geometry_parity.py [case-filter]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARS = ("PG_TNP_C", "PG_TNP_LAG", "PG_TNP_LAGR", "PG_TNP_PF", "PG_TNP_WGS", "PG_TNP_W", "PG_TNP_K1", "PG_TNP_PAIR", "PG_TNP_AHEAD")
CASES = [
    # (row_team.py arguments, geometry C:LAG:LAGR:PF:WGS:W | default).  Defaults (peer_geometry_f32): one wave per column up to 2048
    # rows, two up to 8192, four up to 16384; 16 KiB tiles per wave, LAG = 2 in LDS + LAGR = 2 in registers.
    (["--m", "4096", "--n", "8192"], "default"),                                   # 2 x 2048 rows: W = 1, U = 8, C = 2
    (["--m", "4096", "--n", "8192", "--delay-ns", "0"], "default"),
    (["--m", "4096", "--n", "8192", "--delay-ns", "7000"], "default"),
    (["--m", "4096", "--n", "8192", "--adaptive", "--delay-ns", "3000"], "default"),
    (["--m", "4096", "--n", "8192", "--ranks", "4"], "default"),                   # 4 x 1024 rows: U = 4, C = 4
    (["--m", "4096", "--n", "8192", "--ranks", "8"], "default"),                   # 8 x 512 rows: U = 2, C = 4
    (["--m", "2048", "--n", "4096", "--ranks", "8"], "default"),                   # 8 x 256 rows: U = 1
    (["--m", "16384", "--n", "4096", "--ranks", "8"], "default"),                  # the headline's N = 8 block
    (["--m", "3900", "--n", "1001", "--ranks", "2"], "default"),                   # 1950 rows: U = 8 with a ragged last row group
    (["--m", "3000", "--n", "1001", "--ranks", "2"], "default"),                   # 1500 rows: U = 6
    (["--m", "2305", "--n", "1001", "--ranks", "3"], "default"),                   # 769 / 768 / 768 rows: U = 4 (3 + 1 row groups on rank 0)
    (["--m", "2700", "--n", "513", "--ranks", "2"], "default"),                    # 1350 rows: U = 6 (5.3 row groups)
    (["--m", "2500", "--n", "513", "--ranks", "2"], "default"),                    # 1250 rows: U = 5
    (["--m", "3500", "--n", "513", "--ranks", "2"], "default"),                    # 1750 rows: U = 7
    (["--m", "1400", "--n", "513", "--ranks", "2"], "default"),                    # 700 rows: U = 3
    (["--m", "4096", "--n", "8192", "--fault", "3"], "default"),
    (["--m", "4096", "--n", "8192", "--batched"], "default"),
    (["--m", "4096", "--n", "8192", "--g", "boxv", "--fast", "0"], "default"),
    (["--m", "4096", "--n", "8192", "--g", "l1w"], "default"),
    (["--m", "4096", "--n", "8192", "--then-n", "700"], "default"),
    (["--m", "2048", "--n", "8192", "--dtype", "f64"], "default"),
    (["--m", "4097", "--n", "257"], "default"),                                    # 2049 + 2048 rows: nine row groups -> W = 2, U = 5
    (["--m", "5000", "--n", "257"], "default"),                                    # 2500 rows: W = 2, U = 5
    (["--m", "6100", "--n", "257"], "default"),                                    # 3050 rows: W = 2, U = 6
    (["--m", "7000", "--n", "257"], "default"),                                    # 3500 rows: W = 2, U = 7
    (["--m", "8192", "--n", "4096"], "default"),                                   # 2 x 4096: W = 2, U = 8, C = 2
    (["--m", "8192", "--n", "4096", "--delay-ns", "5000"], "default"),
    (["--m", "10241", "--n", "257"], "default"),                                   # 5121 rows, 21 row groups: W = 2, U = 11
    (["--m", "9000", "--n", "257"], "default"),                                    # U = 9
    (["--m", "10000", "--n", "257"], "default"),                                   # U = 10
    (["--m", "12000", "--n", "257"], "default"),                                   # U = 12
    (["--m", "13000", "--n", "257"], "default"),                                   # U = 13
    (["--m", "14000", "--n", "257"], "default"),                                   # U = 14
    (["--m", "15000", "--n", "257"], "default"),                                   # U = 15
    (["--m", "16384", "--n", "4096"], "default"),                                  # 2 x 8192: W = 2, U = 16
    (["--m", "16384", "--n", "4096", "--delay-ns", "5000"], "default"),
    (["--m", "18433", "--n", "257"], "default"),                                   # 9217 rows, 37 row groups: W = 4, U = 10
    (["--m", "17000", "--n", "257"], "default"),                                   # W = 4, U = 9
    (["--m", "21000", "--n", "257"], "default"),                                   # U = 11
    (["--m", "24000", "--n", "257"], "default"),                                   # U = 12
    (["--m", "26000", "--n", "257"], "default"),                                   # U = 13
    (["--m", "28000", "--n", "257"], "default"),                                   # U = 14
    (["--m", "30000", "--n", "257"], "default"),                                   # U = 15
    (["--m", "32768", "--n", "4096"], "default"),                                  # 2 x 16384 (config 5's block): W = 4, U = 16
    (["--m", "32768", "--n", "4096", "--delay-ns", "6000"], "default"),
    (["--m", "49152", "--n", "1024", "--ranks", "3", "--adaptive"], "default"),
    # Float64: the same table by row groups of 128 rows (two granules per value)
    (["--m", "2048", "--n", "4096", "--dtype", "f64"], "default"),                 # 2 x 1024 rows = 8 row groups: W = 1, U = 8
    (["--m", "2048", "--n", "4096", "--dtype", "f64", "--ranks", "4"], "default"), # 4 row groups: U = 4, C = 2
    (["--m", "1500", "--n", "1001", "--dtype", "f64", "--ranks", "3"], "default"), # 500 rows: U = 4 ragged
    (["--m", "700", "--n", "513", "--dtype", "f64", "--ranks", "2"], "default"),   # 350 rows: U = 3
    (["--m", "400", "--n", "513", "--dtype", "f64", "--ranks", "2"], "default"),   # 200 rows: U = 2
    (["--m", "200", "--n", "513", "--dtype", "f64", "--ranks", "2"], "default"),   # 100 rows: U = 1
    (["--m", "1600", "--n", "513", "--dtype", "f64"], "default"),                  # 800 rows: U = 7
    (["--m", "3000", "--n", "513", "--dtype", "f64"], "default"),                  # 1500 rows, 12 row groups: W = 2, U = 6
    (["--m", "4096", "--n", "4096", "--dtype", "f64"], "default"),                 # 2 x 2048 rows = 16 row groups: W = 2, U = 8
    (["--m", "4096", "--n", "4096", "--dtype", "f64", "--delay-ns", "5000", "--adaptive"], "default"),
    (["--m", "6000", "--n", "513", "--dtype", "f64"], "default"),                  # 24 row groups: W = 2, U = 12
    (["--m", "8192", "--n", "2048", "--dtype", "f64"], "default"),                 # 32 row groups: W = 2, U = 16
    (["--m", "10753", "--n", "257", "--dtype", "f64"], "default"),                 # 5377 rows, 43 row groups: W = 4, U = 11
    (["--m", "16384", "--n", "2048", "--dtype", "f64"], "default"),                # 2 x 8192 rows = 64 row groups: W = 4, U = 16
    (["--m", "16384", "--n", "2048", "--dtype", "f64", "--delay-ns", "6000"], "default"),
    (["--m", "5000", "--n", "1001", "--ranks", "3", "--dtype", "f64", "--g", "boxv", "--fast", "0"], "default"),
    # round 4's geometries (four waves per column, LAG = 2 only): still instantiated for the "before" curve
    (["--m", "4096", "--n", "8192"], "2:2:0:2:4:4"),
    (["--m", "8192", "--n", "4096"], "2:2:0:2:2:4"),
    (["--m", "16384", "--n", "4096"], "1:2:0:2:2:4"),
    (["--m", "32768", "--n", "4096"], "1:2:0:2:1:4"),
    (["--m", "32768", "--n", "4096"], "1:2:1:2:1:4"),
    # round 5's one-wave kernel (gemv_tnt_kernel<..., W = 1>), kept at U = 8 beside round 6's gemv_tnp1_kernel (K1 = 0)
    (["--m", "4096", "--n", "8192"], "2:2:2:2:4:1:0"),
    (["--m", "4096", "--n", "8192", "--delay-ns", "5000"], "2:2:2:2:4:1:0"),
    # round 6's one-wave sweep sums a step's granules over the members by DPP row shifts and, beyond one 16-lane row, permlane swaps:
    # teams whose granules fill two rows (32 lanes) and four (64)
    (["--m", "32768", "--n", "1024", "--ranks", "16"], "default"),                 # 16 x 2048 rows: U = 8, C = 2 -> 32 granules per step
    (["--m", "4096", "--n", "2048", "--ranks", "16"], "default"),                  # 16 x 256 rows: U = 1, C = 4 -> 64 granules
    (["--m", "8192", "--n", "2048", "--ranks", "16", "--adaptive"], "default"),    # 16 x 512 rows: U = 2, C = 4 -> 64
    (["--m", "2048", "--n", "1024", "--dtype", "f64", "--ranks", "8"], "default"), # 8 x 256 rows of Float64: U = 2, C = 2, two granules per value -> 32
    (["--m", "2048", "--n", "1024", "--dtype", "f64", "--ranks", "16"], "default"),# 16 x 128 rows: U = 1 -> 64
    (["--m", "16384", "--n", "1024", "--dtype", "f64", "--ranks", "16", "--g", "l1w"], "default"),  # 16 x 1024 rows: U = 8 -> 64
    # one post per two steps (PAIR) in every shape of the pair's slot: C = 2 / C = 4 / Float64 (8 granules per device), an odd step count, the injector
    (["--m", "4096", "--n", "8192"], "2:2:2:2:4:1:1:1:1"),
    (["--m", "4096", "--n", "8193", "--delay-ns", "5000"], "2:2:2:2:4:1:1:1:1"),
    (["--m", "4096", "--n", "8192", "--ranks", "4", "--tune", "PAIR=1"], "default"),                 # U = 4, C = 4: 8 granules per device and pair
    (["--m", "16384", "--n", "4096", "--ranks", "8", "--tune", "PAIR=1"], "default"),                # 8 x 4 = 32 polling lanes
    (["--m", "32768", "--n", "1024", "--ranks", "16", "--tune", "PAIR=1"], "default"),               # 16 x 4 = 64
    (["--m", "2048", "--n", "4096", "--dtype", "f64", "--tune", "PAIR=1", "--adaptive"], "default"), # Float64: 8 granules per device and pair
    (["--m", "2048", "--n", "1024", "--dtype", "f64", "--ranks", "8", "--tune", "PAIR=1"], "default"),# 8 x 8 = 64
    (["--m", "2048", "--n", "1024", "--dtype", "f64", "--ranks", "16", "--tune", "PAIR=1"], "default"),# 16 devices: the pair does not fit 64 lanes -> one post per step
    (["--m", "4096", "--n", "8192", "--tune", "AHEAD=2", "--g", "l1w"], "default"),
    (["--m", "4096", "--n", "8192", "--g", "box"], "default"),
    (["--m", "4096", "--n", "8192", "--g", "boxv", "--adaptive"], "default"),
]


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    bad = 0
    for args, geom in CASES:
        label = " ".join(args) + " @ " + geom
        if flt and flt not in label:
            continue
        env = dict(os.environ, PG_TUNE="1")
        if geom != "default":
            for v, x in zip(VARS, geom.split(":")):
                env[v] = x
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "row_team.py"), "--steps", "12"] + args,
                             capture_output=True, text=True, timeout=600, env=env)
        try:
            d = json.loads(out.stdout.splitlines()[-1])
        except Exception:  # noqa: BLE001
            print("FAIL", label, "rc", out.returncode, (out.stdout[-300:] + out.stderr[-600:]).replace("\n", " | "))
            bad += 1
            continue
        if "error" in d:
            print("FAIL", label, str(d["error"])[:400].replace("\n", " | "))
            bad += 1
            continue
        dz = max(r["dz"] / r["z_scale"] for rows in d["steps"] for r in rows)
        passes = sorted({r["a_passes"] for rows in d["steps"] for r in rows if r["k"] >= 2})
        flags = sorted({r["flags"] & d["fallback_flag"] for rows in d["steps"] for r in rows})
        gam = all(abs(r["gamma"] - r["gamma_oracle"]) <= 1e-6 * abs(r["gamma_oracle"]) for rows in d["steps"] for r in rows)
        special = "--fault" in args
        ok = dz <= (1e-11 if "f64" in args else 1e-5) and d["ranks_agree_bitwise"] and gam and all(v == "ok" for v in d["selftest"]) and \
            (special or (passes == [1] and flags == [0]))
        if "--batched" in args:
            ok = ok and all(bt["k"] == 13 and bt["dz_rel"] <= 1e-5 for bt in d["batched"])
        bad += 0 if ok else 1
        print("ok  " if ok else "FAIL", label, "max dz/scale %.2e" % dz, "agree", d["ranks_agree_bitwise"], "passes", passes, "fallback flags", flags,
              "gamma", gam, flush=True)
    print("bad cases:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
