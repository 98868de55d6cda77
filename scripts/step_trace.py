#!/usr/bin/env python3
"""What one iteration launches beside its sweep: python scripts/step_trace.py "<dir>/*/*_results.db" [sweep-name-substring]
From a rocprofv3 --kernel-trace database, takes the steady-state steps (the kernels from one sweep launch up to the next)
and prints the launch sequence of the median step with each kernel's duration and the idle gap in front of it, then the
step's totals: sweep time, time in the other kernels, idle time, number of launches beside the sweep."""
import glob
import re
import sqlite3
import statistics
import sys

db = sorted(glob.glob(sys.argv[1]))[-1]
key = sys.argv[2] if len(sys.argv) > 2 else "gemv_tn"
con = sqlite3.connect(db)
rows = con.execute("select name, start, end from kernels order by start").fetchall()


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n[:90]


idx = [i for i, r in enumerate(rows) if key in r[0]]
steps = []
for a, b in zip(idx, idx[1:]):
    seg = rows[a:b + 1]
    span = (seg[-1][1] - seg[0][1]) / 1e3
    steps.append((span, a, b))
if len(steps) < 5:
    sys.exit("fewer than 5 steps with a '%s' kernel" % key)
# the timed region: steps whose launch count equals the most common one among the last two thirds
tail = steps[len(steps) // 3:]
counts = [b - a for _, a, b in tail]
mode = statistics.mode(counts)
tail = [s for s in tail if s[2] - s[1] == mode]
tail.sort()
span, a, b = tail[len(tail) // 2]
print("median steady-state step of %d (db %s): %.1f us from sweep start to next sweep start, %d launches beside the sweep\n" % (
    len(tail), db.split("/")[-1], span, mode - 1))
print("| # | kernel | gap before us | duration us |\n|---:|---|---:|---:|")
busy_other = idle = 0.0
for k in range(a, b):
    n, s, e = rows[k]
    gap = (s - rows[k - 1][2]) / 1e3 if k > a else 0.0
    print("| %d | `%s` | %.1f | %.1f |" % (k - a, short(n), gap, (e - s) / 1e3))
    if k > a:
        busy_other += (e - s) / 1e3
        idle += max(gap, 0.0)
idle += max((rows[b][1] - rows[b - 1][2]) / 1e3, 0.0)
sweep = (rows[a][2] - rows[a][1]) / 1e3
print("\nsweep %.1f us | other kernels %.1f us | idle %.1f us | step overhead beside the sweep %.1f us (%.2f %% of the step)" % (
    sweep, busy_other, idle, span - sweep, 100 * (span - sweep) / span))
med = lambda f: statistics.median(f(s) for s in tail)
print("median over the %d steps: step %.1f us, overhead %.1f us" % (
    len(tail), med(lambda s: s[0]), med(lambda s: s[0] - (rows[s[1]][2] - rows[s[1]][1]) / 1e3)))
