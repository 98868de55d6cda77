#!/usr/bin/env python3
"""Tabulate tests/tools/row_team_sweep.py output (JSON lines): one row per geometry, one column per injected latency; the figure is
bytes of A per ITERATION / time in TB/s (all ranks together), `*` = the run fell back to two sweeps.
    python scripts/r5_sweep_table.py gpurun_out/r5b/sweep_2048.jsonl [...] [--md]"""
import collections
import json
import sys

md = "--md" in sys.argv
for path in [a for a in sys.argv[1:] if not a.startswith("--")]:
    rows = [json.loads(l) for l in open(path) if l.startswith("{")]
    rows = [r for r in rows if "error" not in r or not print("ERR", r)]
    if not rows:
        continue
    tab = collections.OrderedDict()
    for r in rows:
        tab.setdefault(r["geometry"], collections.OrderedDict()).setdefault(r["delay_ns"], []).append(r)
    delays = []
    for t in tab.values():
        for d in t:
            if d not in delays:
                delays.append(d)
    r0 = rows[0]
    es = 4 if r0["dtype"] == "f32" else 8
    print("%s## %d ranks x %d rows, n = %d (%s)" % ("" if md else "", r0["ranks"], r0["m"] // r0["ranks"], r0["n"], path))
    head = ["geometry C:LAG:LAGR:PF:WGS:OPT"] + ["off" if d is None else "%g us" % (d / 1000) for d in delays] + ["late % (off)", "slack us (0)"]
    if md:
        print("| " + " | ".join(head) + " |\n|" + "---|" * len(head))
    else:
        print("%-20s" % head[0] + "".join("%9s" % h for h in head[1:-2]) + "   late%  slack")
    for g, t in tab.items():
        cells = []
        for d in delays:
            if d in t:
                vs = [r["m"] * r["n"] * es * r["it_per_s"] / 1e12 for r in t[d]]
                star = "*" if any(r["a_passes_per_step"] > 1.5 for r in t[d]) else ""
                cells.append("/".join("%.2f" % v for v in vs) + star)
            else:
                cells.append("-")
        roff = (t.get(None) or [None])[0]
        late = ""
        if roff and g[0].isdigit():
            C = int(g.split(":")[0])
            wave_steps = roff["n"] / C * 4 * roff["steps"]
            late = "%.0f" % (100.0 * sum(roff["late_waves"]) / (roff["ranks"] * wave_steps))
        r00 = (t.get(0) or [None])[0]
        slack = "" if not r00 or not r00.get("slack_us") or r00["slack_us"][0] is None else "%.1f" % (sum(r00["slack_us"]) / len(r00["slack_us"]))
        if md:
            print("| " + " | ".join([g] + cells + [late, slack]) + " |")
        else:
            print("%-20s" % g + "".join("%9s" % c for c in cells) + "   %5s  %5s" % (late, slack))
    print()
