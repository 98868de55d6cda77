#!/usr/bin/env python3
"""Parity of row-team sweep geometries chosen through the tuning variables (round 5: lag tiles in registers, latency injector):
runs tests/tools/row_team.py (iterates of every rank against the CPU restatement on the whole matrix) once per case and prints
one summary line each.   python scripts/r5_peer_geometry_parity.py [case-filter]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARS = ("PG_TNP_C", "PG_TNP_LAG", "PG_TNP_LAGR", "PG_TNP_PF", "PG_TNP_WGS", "PG_TNP_W")
CASES = [
    # (row_team.py arguments, geometry C:LAG:LAGR:PF:WGS:W)
    (["--m", "4096", "--n", "8192"], "default"),
    (["--m", "4096", "--n", "8192"], "2:3:3:2:3:4"),
    (["--m", "4096", "--n", "8192"], "4:2:2:2:2:4"),
    (["--m", "4096", "--n", "8192"], "2:2:0:2:4:1"),
    (["--m", "4096", "--n", "8192"], "2:2:1:2:4:1"),
    (["--m", "4096", "--n", "8192"], "2:2:2:2:4:1"),
    (["--m", "4096", "--n", "8192"], "2:2:3:2:4:1"),
    (["--m", "4096", "--n", "8192"], "2:2:2:1:4:1"),
    (["--m", "4096", "--n", "8192", "--delay-ns", "0"], "2:2:2:2:4:1"),
    (["--m", "4096", "--n", "8192", "--delay-ns", "7000"], "2:2:2:2:4:1"),
    (["--m", "4096", "--n", "8192", "--adaptive", "--delay-ns", "3000"], "2:2:2:2:4:1"),
    (["--m", "4096", "--n", "8192", "--ranks", "4"], "2:2:2:2:4:1"),
    (["--m", "16384", "--n", "4096", "--ranks", "8"], "2:2:2:2:4:1"),
    (["--m", "3900", "--n", "1001", "--ranks", "2"], "2:2:2:2:4:1"),
    (["--m", "3000", "--n", "1001", "--ranks", "2"], "2:2:2:2:4:1"),
    (["--m", "4096", "--n", "8192", "--fault", "3"], "2:2:2:2:4:1"),
    (["--m", "4096", "--n", "8192", "--batched"], "2:2:2:2:4:1"),
    (["--m", "4096", "--n", "8192", "--g", "boxv", "--fast", "0"], "2:2:2:2:4:1"),
    (["--m", "4096", "--n", "8192", "--then-n", "700"], "2:2:2:2:4:1"),
    (["--m", "2048", "--n", "8192", "--dtype", "f64"], "default"),
    (["--m", "8192", "--n", "4096"], "default"),
    (["--m", "8192", "--n", "4096"], "2:2:1:2:2:4"),
    (["--m", "8192", "--n", "4096"], "2:2:2:2:2:4"),
    (["--m", "8192", "--n", "4096"], "1:2:0:2:4:1"),
    (["--m", "8192", "--n", "4096"], "1:2:1:2:4:1"),
    (["--m", "8192", "--n", "4096"], "1:2:2:2:4:1"),
    (["--m", "8192", "--n", "4096"], "2:2:0:2:2:2"),
    (["--m", "8192", "--n", "4096"], "2:2:2:2:2:2"),
    (["--m", "16384", "--n", "4096"], "default"),
    (["--m", "16384", "--n", "4096"], "1:2:1:2:2:4"),
    (["--m", "16384", "--n", "4096"], "1:2:0:2:2:2"),
    (["--m", "16384", "--n", "4096"], "1:2:1:2:2:2"),
    (["--m", "16384", "--n", "4096"], "1:2:2:2:2:2"),
    (["--m", "32768", "--n", "4096"], "default"),
    (["--m", "32768", "--n", "4096"], "1:2:1:2:1:4"),
    (["--m", "32768", "--n", "4096"], "1:2:2:2:1:4"),
    (["--m", "32768", "--n", "4096"], "1:0:2:2:1:4"),
    (["--m", "32768", "--n", "4096", "--delay-ns", "6000"], "1:2:2:2:1:4"),
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
