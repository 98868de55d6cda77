#!/usr/bin/env python3
"""Randomised runs of the PRODUCTION multi-rank path on one GPU: `bench.py --gpus N --share-device --backend gloo` (one process per
rank, torch.distributed rendezvous; for row teams the inboxes exported, all-gathered and imported through IPC handles) against
`bench.py --gpus 1` on the same synthetic problem: same lambda / Lf, and after the same number of iterations the same gamma,
f(x), g(z) and stopping measure.  Random shapes (ragged shards), rank counts 2..4, layouts rows / rows as a row team / cols,
fixed / adaptive step, f32 / f64.  Usage: python tests/tools/fuzz_bench_ranks.py [cases] [first_seed]."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, nproc, port):
    common = ["--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-also", "--sustain", "0"] + args
    if nproc == 1:
        cmd = [sys.executable, BENCH] + common
    else:  # bench.py starts its own ranks (the way the driver's `python bench.py --gpus N` does) and forwards the script's options
        cmd = [sys.executable, BENCH, "--gpus", str(nproc), "--share-device", "--backend", "gloo", "--no-row-teams"] + common
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        return None, "exit %d: %s" % (out.returncode, (out.stdout[-400:] + out.stderr[-800:]).replace("\n", " | "))
    lines = out.stdout.splitlines()
    if len(lines) != 1:
        return None, "stdout has %d lines" % len(lines)
    return json.loads(lines[0]), ""


def close(a, b, rel):
    return abs(a - b) <= rel * max(abs(a), abs(b), 1e-30)


def one_case(seed):
    rng = np.random.default_rng(seed)
    f64 = bool(rng.random() < 0.25)
    world = int(rng.choice([2, 2, 3, 4]))
    layout = str(rng.choice(["rows", "teams", "teams", "cols"]))
    mode = str(rng.choice(["fixed", "adaptive"]))
    if layout == "cols":
        m = int(rng.choice([512, 2048, 4100, 8192, 20000, 40000]))
        n = int(rng.integers(world * 40, 6000))
    else:
        cap = (8192 if f64 else 16384) * world
        m = int(rng.choice([world * 300, world * 2048, world * 2048 + 1, world * 4096 - 1, 5000, 12000, cap]))
        m = min(m, cap)
        n = int(rng.integers(64, 6000))
    return run_pair(world, layout, m, n, mode, f64, "seed=%d " % seed)


def run_pair(world, layout, m, n, mode, f64, tag=""):
    """one problem as one rank and as `world` ranks (processes sharing the device) in `layout`; ("", label) when they agree"""
    args = ["--m", str(m), "--n", str(n), "--mode", mode] + (["--dtype", "f64"] if f64 else [])
    label = "%sworld=%d layout=%s %s" % (tag, world, layout, " ".join(args))
    one, why = run(args, 1, 0)
    if one is None:
        return "single rank: " + why, label
    extra = {"rows": ["--sharding", "rows"], "teams": ["--sharding", "rows", "--row-teams"], "cols": ["--sharding", "cols"]}[layout]
    many, why = run(args + extra, world, 0)
    if many is None:
        return "%d ranks: %s" % (world, why), label
    c1, c2 = one["config"], many["config"]
    rel = 1e-9 if f64 else 3e-4
    l1, l2 = c1["problem"]["lambda"], c2["problem"]["lambda"]
    if not close(l1, l2, 1e-5 if not f64 else 1e-12):
        return "lambda %r / %r" % (l1, l2), label
    f1, f2 = c1["final"], c2["final"]
    # (the stopping measure is a difference of nearly equal vectors: near convergence it sits on the rounding floor of the element type)
    floor = {"res_inf_over_gamma": 1e-11 if f64 else 3e-6}
    for key, r in (("gamma", rel), ("f_x", rel), ("g_z", rel), ("res_inf_over_gamma", 30 * rel)):
        if not close(f1[key], f2[key], r) and abs(f1[key] - f2[key]) > floor.get(key, 0.0):
            return "%s after the same iterations: %r (one rank) / %r (%d ranks)" % (key, f1[key], f2[key], world), label
    if layout == "teams":
        if not (c2.get("row_teams") and c2.get("row_team_selftest") == "ok"):
            return "row team not formed: %r" % ({k: c2.get(k) for k in ("row_teams", "row_team_selftest")},), label
        if c2.get("sweep_fallbacks") or (mode == "fixed" and abs(c2["a_passes_per_step"] - 1.0) > 0.1):
            return "row team: %r fallbacks, %r reads per step" % (c2.get("sweep_fallbacks"), c2["a_passes_per_step"]), label
    if layout == "cols" and abs(c2["a_passes_per_step"] - 1.0) > 0.1 and mode == "fixed":
        return "column shards: %r reads per step" % (c2["a_passes_per_step"],), label
    return "", label


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time()
    bad = 0
    for seed in range(seed0, seed0 + cases):
        try:
            why, label = one_case(seed)
        except Exception as e:  # noqa: BLE001
            why, label = "%s: %s" % (type(e).__name__, e), "seed=%d" % seed
        if why:
            bad += 1
            print("FAIL", label, "--", why, flush=True)
    print("%d cases, %d failing, %.1f s" % (cases, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
