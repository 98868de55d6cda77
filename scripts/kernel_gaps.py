#!/usr/bin/env python3
"""Inter-kernel gaps from a rocprofv3 --kernel-trace database: python scripts/kernel_gaps.py "<dir>/*/*_results.db".
Reports the idle time between the epilogue of one FFB iteration and the extrapolation of the next (host round trip) and
the gaps between the kernels of an iteration (tracing itself inflates them)."""
import sqlite3, sys, glob
db = sorted(glob.glob(sys.argv[1]))[-1]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
st = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = con.execute(f"select s.kernel_name, k.start, k.end from {kt} k join {st} s on k.kernel_id = s.id order by k.start").fetchall()
# gaps between an EpilogueF kernel end and the following kernel start, and all inter-kernel gaps inside an iteration
import statistics
gaps_iter, gaps_all = [], []
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    g = (s1 - e0) / 1e3
    if "EpilogueF" in n0 and "ExtrapolateF" in n1: gaps_iter.append(g)
    elif g < 1000: gaps_all.append(g)
print("iteration boundary gap (epilogue end -> next extrapolate start), us: n=%d median=%.1f min=%.1f max=%.1f" % (len(gaps_iter), statistics.median(gaps_iter), min(gaps_iter), max(gaps_iter)))
print("other inter-kernel gaps, us: n=%d median=%.1f mean=%.1f" % (len(gaps_all), statistics.median(gaps_all), sum(gaps_all)/len(gaps_all)))
