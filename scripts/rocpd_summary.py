#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd SQLite) result: per-kernel launch statistics and, when PMC counters were
collected, per-kernel counter averages.  Usage: rocpd_summary.py [--sum-per-dispatch] [--match SUBSTR] results.db [more.db ...] > summary.md
--sum-per-dispatch adds the per-shader-engine / per-channel samples of one dispatch before averaging over dispatches."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= 110 else name[:107] + "..."


def summarise(path, match=None, per_dispatch=False):
    con = sqlite3.connect(path)
    cur = con.cursor()
    print(f"## {path}\n")
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
        "group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, n, tot, avg, mn, mx in rows:
        if match and match not in name:
            continue
        print(f"| `{short(name)}` | {n} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.2f} |")
    if per_dispatch:
        # SQ / TCC counters are sampled per shader engine / channel: sum the samples of one dispatch, then average over dispatches
        try:
            pmc = cur.execute(
                "select name, counter_name, count(*), avg(v), min(v), max(v) from (select name, counter_name, dispatch_id, "
                "sum(counter_value) as v from pmc_events group by name, counter_name, dispatch_id) "
                "group by name, counter_name order by name, counter_name").fetchall()
        except sqlite3.OperationalError as e:
            print(f"(per-dispatch sums unavailable: {e})")
            pmc = []
    else:
        pmc = cur.execute(
            "select name, counter_name, count(*), avg(counter_value), min(counter_value), max(counter_value) "
            "from pmc_events group by name, counter_name order by avg(counter_value) desc").fetchall()
    if match:
        pmc = [r for r in pmc if match in r[0]]
    if pmc:
        print("\n| kernel | counter | samples | avg | min | max |")
        print("|---|---|---:|---:|---:|---:|")
        for name, cname, n, avg, mn, mx in pmc:
            print(f"| `{short(name)}` | {cname} | {n} | {avg:.6g} | {mn:.6g} | {mx:.6g} |")
    print()


if __name__ == "__main__":
    argv = sys.argv[1:]
    per_dispatch = "--sum-per-dispatch" in argv
    argv = [a for a in argv if a != "--sum-per-dispatch"]
    match = None
    if "--match" in argv:
        i = argv.index("--match")
        match = argv[i + 1]
        del argv[i:i + 2]
    for p in argv:
        summarise(p, match, per_dispatch)
