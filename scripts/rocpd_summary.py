#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd SQLite) result: per-kernel launch statistics and, when PMC counters were
collected, per-kernel counter averages.  Usage: rocpd_summary.py results.db [more.db ...] > summary.md"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= 110 else name[:107] + "..."


def summarise(path):
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
        print(f"| `{short(name)}` | {n} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.2f} |")
    pmc = cur.execute(
        "select name, counter_name, count(*), avg(counter_value), min(counter_value), max(counter_value) "
        "from pmc_events group by name, counter_name order by avg(counter_value) desc").fetchall()
    if pmc:
        print("\n| kernel | counter | samples | avg | min | max |")
        print("|---|---|---:|---:|---:|---:|")
        for name, cname, n, avg, mn, mx in pmc:
            print(f"| `{short(name)}` | {cname} | {n} | {avg:.6g} | {mn:.6g} | {mx:.6g} |")
    print()


if __name__ == "__main__":
    for p in sys.argv[1:]:
        summarise(p)
