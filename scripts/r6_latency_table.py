#!/usr/bin/env python3
"""profiles/r6_row_team_latency_sweep.md from the lines tests/tools/row_team_sweep.py left under gpurun_out/r6/sweep_*.jsonl
(scripts/collect_round6_profiles.sh latency).    python scripts/r6_latency_table.py [dir] > profiles/r6_row_team_latency_sweep.md"""
import collections
import json
import os
import sys

D = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r6")
NAMES = {"2:2:2:2:4:1:0": "round 5 (`gemv_tnt<8,2,1,2,2,PEER,2>`)", "default": "**round 6 default**", "2:2:2:2:4:1:1:0:0": "round 6, poll in its own step (`AHEAD=2`)",
         "2:2:2:2:4:1:1:1:1": "round 6, one post per two steps (`PAIR=1`)", "two-sweeps": "two sweeps + all-reduce (reads A twice)"}
FILES = [("sweep_2048", "2 ranks x 2048 rows, n = 1048576 (the headline's block at N = 8)"), ("sweep_4096", "2 ranks x 4096 rows, n = 524288 (two waves per column)"),
         ("sweep_16384", "2 ranks x 16384 rows, n = 131072 (config 5's block; four waves per column)"), ("sweep_4x4096", "4 ranks x 4096 rows, n = 262144"),
         ("sweep_8x2048", "8 ranks x 2048 rows, n = 262144 (eight contexts on one device: 128 workgroups each)"), ("sweep_f64_1024", "Float64: 2 ranks x 1024 rows, n = 1048576")]


def table(path):
    rows = collections.OrderedDict()
    meta = {}
    for ln in open(path):
        try:
            d = json.loads(ln)
        except ValueError:
            continue
        if "error" in d:
            continue
        g, dl = d["geometry"], "off" if d["delay_ns"] is None else "%g" % (d["delay_ns"] / 1000.0)
        rows.setdefault(g, collections.OrderedDict()).setdefault(dl, []).append(d["TBps_all_ranks"] * (0.5 if g == "two-sweeps" else 1.0))
        m = meta.setdefault(g, {})
        # wave-steps of the timed sweeps: every column group is visited once by the W waves that share a column (peer_geometry)
        rg = -(-(d["m"] // d["ranks"]) // (256 if d["dtype"] == "f32" else 128))
        W = 1 if rg <= 8 else (2 if rg <= 32 else 4)
        U = -(-rg // W)
        C = (4 if (U <= 4 and d["dtype"] == "f32") else 2) if U <= 8 else 1
        steps = d["steps"] * -(-d["n"] // C) * W
        if dl == "off":
            m["late"] = 100.0 * max(d["late_waves"]) / max(steps, 1)
        if dl == "0" and d["slack_us"] and d["slack_us"][0]:
            m["slack"] = sum(d["slack_us"]) / len(d["slack_us"])
    cols = []
    for g in rows:
        for dl in rows[g]:
            if dl not in cols:
                cols.append(dl)
    out = ["| sweep | " + " | ".join(c if c == "off" else c + " us" for c in cols) + " | late % (off) | slack us (0) |", "|---|" + "---|" * (len(cols) + 2)]
    for g, dd in rows.items():
        cells = ["/".join("%.2f" % v for v in dd[c]) if c in dd else "-" for c in cols]
        m = meta.get(g, {})
        out.append("| %s | " % NAMES.get(g, "`%s`" % g) + " | ".join(cells) + " | %s | %s |" % ("%.1f" % m["late"] if "late" in m and g != "two-sweeps" else "", "%.1f" % m["slack"] if "slack" in m else ""))
    return "\n".join(out)


def main():
    print("""# Row-team sweep under injected hand-off latency, round 6 (VERDICT r5 next-round 1: "still flat through 8 us")

What is measured: as `profiles/r5_row_team_latency_sweep.md` -- the ranks of a row team share ONE MI355X as contexts of one process
(`tests/tools/row_team_sweep.py`: row blocks generated once, every sweep form and latency timed on them, 20 FastForwardBackward iterations
after 4 warm-up steps, fixed step, L1).  **Every cell: bytes of A per ITERATION / time in TB/s, all ranks together** (8 TB/s = the chip's HBM
peak; the `two sweeps` row: half of the bytes it reads).  The injector (`pg_ctx_test_team_fault(ctx, ns, 2)`): next to its granules a member
posts a stamp of the device's 100 MHz clock, a consumer accepts a step's granules only once every member's stamp is `ns` old -- the x axis is
the TOTAL latency of the hop, max(on-chip hand-off, injected).  `off` = the product kernel without the injector.  `late %` = wave-steps that
did not find their granules at the first look (injector off, the worst rank); `slack` = mean time between a granule's store and its use at
0 us injected.  Two numbers in a cell = two rounds.  ONE box for the whole file (`scripts/collect_round6_profiles.sh latency`); boxes of this
pool differ by +-5 %, so round 5's kernel is in the same table.
""")
    for name, title in FILES:
        path = os.path.join(D, name + ".jsonl")
        if os.path.exists(path):
            print("### %s\n" % title)
            print(table(path))
            print()


if __name__ == "__main__":
    main()
