#!/usr/bin/env python3
"""Which earlier test of tests/test_gpu_parity.py makes test_bench_default_line_carries_every_single_gpu_config fail when it runs in
the same pytest process before it?  Binary search over the tests that precede it in collection order (GPU box)."""
import subprocess
import sys

TARGET = "test_bench_default_line_carries_every_single_gpu_config"


def collect():
    out = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-m", "gpu", "--collect-only", "-q"], capture_output=True, text=True).stdout
    ids = [l.strip() for l in out.splitlines() if "::" in l]
    k = [i for i, t in enumerate(ids) if TARGET in t][0]
    return ids[:k], ids[k]


def fails(subset, target):
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider"] + subset + [target], capture_output=True, text=True)
    tail = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else ""
    bad = TARGET in out.stdout and "FAILED" in out.stdout and ("FAILED " + target.split("::")[0]) in out.stdout
    bad = any(TARGET in l and l.startswith("FAILED") for l in out.stdout.splitlines())
    why = [l for l in out.stdout.splitlines() if "AssertionError: ([" in l]
    print("  %4d tests before it -> %s   %s %s" % (len(subset), "FAILS" if bad else "passes", tail, (why[0][:160] if why else "")), flush=True)
    return bad


def main():
    before, target = collect()
    print(len(before), "tests precede", target)
    if not fails(before, target):
        print("the full prefix passes: not reproducible in this file alone")
        return
    cand = before
    while len(cand) > 1:
        h = len(cand) // 2
        a, b = cand[:h], cand[h:]
        if fails(a, target):
            cand = a
        elif fails(b, target):
            cand = b
        else:
            print("needs tests from both halves:", len(a), len(b))
            break
    print("culprit(s):")
    for t in cand[:20]:
        print("  ", t)


if __name__ == "__main__":
    main()
