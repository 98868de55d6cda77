#!/usr/bin/env python3
"""Randomised runs of tests/tools/row_team.py (north_star's row layout at one read of A per iteration, the ranks as contexts of
one process on one GPU): random rank counts, block lengths (every PEER geometry: U = 2 .. 16, ragged last row groups, blocks of
different length on different ranks), column counts, element types, FB / FFB, fixed / adaptive step, L1 / box, the batched
in-library loop, a second problem on the same contexts.  Per case the
checks of test_row_team_iterates_match_oracle_at_one_read_of_A: every rank's iterates equal the CPU restatement on the WHOLE
matrix, the ranks agree bit for bit, the self-test passed, and from the second step on a step that is not flagged as a
fallback is ONE read of the row block.  Usage: python tests/tools/fuzz_row_team.py [cases] [first_seed] [cols].  `cols`: the
same campaign over COLUMN shards (rank p holds A[:, J_p]; every column length incl. the team sweep's; the ranks' slices against
the restatement's, the iteration's scalars bit-identical on all ranks)."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


COLS = False
MANY = False  # `many` on the command line: 6 / 7 / 12 / 16 ranks (16 is the most a row team takes)
PLAIN = False  # `plain` on the command line: row shards WITHOUT the team (two sweeps + the all-reduce of [grad ; f]: north_star's layout as rounds 1-3 ran it)
PER_ELEMENT = False  # `pe` on the command line: per-element weights of NormL1 / bounds of IndBox instead of scalars


def draw_cols(seed):
    """column shards (the default N > 1 layout of bench.py): rank p holds A[:, J_p]; every column length / sweep geometry, n >= ranks"""
    rng = np.random.default_rng(seed)
    f64 = bool(rng.random() < 0.3)
    ranks = int(rng.choice([2, 2, 3, 4, 5, 8]))
    if rng.random() < 0.5:
        m = int(rng.choice([1, 100, 256, 2048, 2305, 4096, 7000, 8192, 16384, 20000, 33000, 40000, 70001, 131072]))
    else:
        m = int(rng.integers(1, 150000))
    if f64:
        m = min(m, 131072)
    n = int(rng.choice([8, 9, 31, 64, 65, 500, 1001])) if rng.random() < 0.5 else int(rng.integers(ranks, 1500))
    n = max(n, ranks)
    while m * n > 2.5e7 and n // 2 >= ranks:
        n = n // 2
    args = ["--cols", "--m", str(m), "--n", str(n), "--ranks", str(ranks), "--steps", "8"]
    if f64:
        args += ["--dtype", "f64"]
    fast = not (rng.random() < 0.35)
    if not fast:
        args += ["--fast", "0"]
    adaptive = bool(fast and rng.random() < 0.35)  # (adaptive step under column shards: FastForwardBackward with the residual pair)
    if adaptive:
        args += ["--adaptive"]
    if rng.random() < 0.3:
        args += ["--g", "box"]
    if fast and not adaptive and rng.random() < 0.3:
        args += ["--batched"]
    if PER_ELEMENT:
        args = [a for a in args if a not in ("--g", "box")] + ["--g", str(np.random.default_rng(seed + 7_000_003).choice(["l1w", "boxv"]))]
    return args, (1e-11 if f64 else 1e-5), adaptive


def draw(seed):
    if COLS:
        return draw_cols(seed)
    rng = np.random.default_rng(seed)
    f64 = bool(rng.random() < 0.3)
    ranks = int(rng.choice([2, 2, 3, 4, 5, 8]))
    if MANY:
        ranks = int(np.random.default_rng(seed + 9_000_011).choice([6, 7, 12, 16]))
    cap = 8192 if f64 else 16384  # rows per rank the PEER sweep covers
    if rng.random() < 0.5:
        rpr = int(rng.choice([1, 100, 256, 257, 512, 1024, 2048, 2049, 4096, 5000, 8192, 12000, 16384]))
    else:
        rpr = int(rng.integers(1, cap + 1))
    rpr = min(rpr, cap)
    m = ranks * rpr - int(rng.integers(0, min(ranks, rpr)))  # ragged: the last ranks hold one row less
    n = int(rng.choice([1, 2, 3, 31, 64, 65, 500, 1001])) if rng.random() < 0.5 else int(rng.integers(1, 1500))
    while m * n > 2.5e7:  # the oracle runs on the whole matrix on the CPU
        n = max(1, n // 2)
    args = ["--m", str(m), "--n", str(n), "--ranks", str(ranks), "--steps", "8"]
    if f64:
        args += ["--dtype", "f64"]
    if rng.random() < 0.35:
        args += ["--fast", "0"]
    adaptive = bool(rng.random() < 0.35)
    if adaptive:
        args += ["--adaptive"]
    if rng.random() < 0.3:
        args += ["--g", "box"]
    # drawn last, from a generator of their own (the cases of the first campaigns keep their seeds): the in-library batched loop
    # afterwards (FastForwardBackward, fixed step), a second problem on the same contexts (another ring layout)
    rx = np.random.default_rng(seed + 7_000_003)
    if PER_ELEMENT:
        args = [a for a in args if a not in ("--g", "box")] + ["--g", str(rx.choice(["l1w", "boxv"]))]
    if "--fast" not in args and not adaptive and rx.random() < 0.3:
        args += ["--batched"]
    if rx.random() < 0.3:
        args += ["--then-n", str(int(rx.integers(1, 900)))]
    return args, (1e-11 if f64 else 1e-5), adaptive


def one_case(seed):
    args, tol, adaptive = draw(seed)
    if PLAIN:
        args = [a for a in args if a != "--batched"] + ["--no-team"]
        if "--then-n" in args:
            i = args.index("--then-n")
            del args[i:i + 2]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "row_team.py")] + args, capture_output=True, text=True,
                         timeout=600)
    label = "seed=%d %s" % (seed, " ".join(args))
    if out.returncode != 0:
        return "exit %d: %s" % (out.returncode, (out.stdout[-600:] + out.stderr[-600:]).replace("\n", " | ")), label, 0
    d = json.loads(out.stdout.splitlines()[-1])
    if not d["ranks_agree_bitwise"]:
        return "the ranks' iterates differ", label, 0
    if not COLS and not PLAIN and not all(v == "ok" for v in d["selftest"]):
        return "self-test: %r" % (d["selftest"],), label, 0
    fallbacks = 0
    for rows in d["steps"]:
        flagged = {r["k"] for r in rows if r["flags"] & d["fallback_flag"]}
        fallbacks += len(flagged)
        for r in rows:
            if not r["dz"] <= tol * r["z_scale"]:
                return "iterate %d off by %.3g (scale %.3g)" % (r["k"], r["dz"], r["z_scale"]), label, fallbacks
            if adaptive and abs(r["gamma"] - r["gamma_oracle"]) > 1e-6 * abs(r["gamma_oracle"]):
                return "gamma %r against the oracle's %r at iteration %d" % (r["gamma"], r["gamma_oracle"], r["k"]), label, fallbacks
            # one read of the block per step from the second step on -- except a step whose sweep was lost (flagged: redone with
            # two sweeps) and the step after it (no speculative half to continue from)
            if PLAIN:
                continue  # (two reads per step by design)
            if r["k"] >= 2 and r["k"] not in flagged and r["k"] - 1 not in flagged and r["a_passes"] != 1 and not adaptive:
                return "iteration %d read the block %d times without a fallback flag; reads by iteration and rank: %r" % (
                    r["k"], r["a_passes"], [[x["a_passes"] for x in rr] for rr in d["steps"]]), label, fallbacks
    if "--batched" in args:
        for bt in d["batched"]:
            # (k < 9: the iterates reached an exact fixed point -- res == 0 stops the loop at tol = 0; the answer is then the oracle's)
            if not (bt["k"] <= 9 and bt["dz_rel"] <= tol and (bt["k"] == 9 or bt["dz_rel"] == 0.0)):
                return "batched loop: %r" % (bt,), label, fallbacks
    if "--then-n" in args:
        for sec in d["second"]:
            if not (sec["max_dz_rel"] <= tol and sec["fallbacks"] == 0):
                return "second problem on the same contexts: %r" % (sec,), label, fallbacks
    if fallbacks:
        print("NOTE", label, "-- %d rank-steps redone with two sweeps: iterations %r" % (
            fallbacks, sorted({r["k"] for rows in d["steps"] for r in rows if r["flags"] & d["fallback_flag"]})), flush=True)
    return "", label, fallbacks


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    global COLS
    global PER_ELEMENT
    COLS = "cols" in sys.argv[3:]
    PER_ELEMENT = "pe" in sys.argv[3:]
    global MANY
    MANY = "many" in sys.argv[3:]
    global PLAIN
    PLAIN = "plain" in sys.argv[3:]
    t0 = time.time()
    bad = fb = 0
    for seed in range(seed0, seed0 + cases):
        try:
            why, label, fallbacks = one_case(seed)
        except Exception as e:  # noqa: BLE001
            why, label, fallbacks = "%s: %s" % (type(e).__name__, e), "seed=%d" % seed, 0
        fb += fallbacks
        if why:
            bad += 1
            print("FAIL", label, "--", why, flush=True)
    print("%d cases, %d failing, %d rank-steps redone with two sweeps, %.1f s" % (cases, bad, fb, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
