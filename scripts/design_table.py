#!/usr/bin/env python3
"""Print the rows of DESIGN.md's "Measured (round 2)" table from the bench lines under profiles/ (run after
scripts/publish_round_profiles.sh), so that the table is transcribed by a program and not by hand."""
import json
import os

P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def line(name):
    for l in open(os.path.join(P, name)):
        if l.startswith("{"):
            return json.loads(l)
    raise SystemExit(name + ": no JSON line")


def ffb(d):
    r = d["roofline"]
    return f"{d['value']:.1f} it/s, {d['ms_per_step']:.3f} ms/step, kernel {r['avg_launch_ms']:.3f} ms, {r['achieved'] / 1e3:.2f} TB/s, frac {r['frac']:.3f}"


d = line("r2_bench_default.json")
print("headline fixed      :", ffb(d), "| traffic", d["roofline"]["traffic"], "| cpu", round(d["cpu_baseline"]["value"], 2), round(d["cpu_baseline"]["value_1thread"], 2))
for a in d["also"]:
    if a["label"] == "config3":
        s, l = a["stepping"], a["device_loop"]
        print(f"also config3        : stepping {s['value']:.0f} it/s, dr_step {s['roofline']['avg_launch_ms'] * 1e3:.1f} us, {s['roofline']['achieved'] / 1e3:.2f} TB/s, frac {s['roofline']['frac']:.3f};"
              f" loop {l['value']:.0f} it/s, dr_block {l['roofline']['avg_launch_ms'] * 1e3:.0f} us")
    elif a["label"] == "config4":
        pk = a["roofline"]["per_kernel"]
        print(f"also config4        : {a['value']:.1f} it/s, " + ", ".join(f"{k} {v['avg_ms']:.2f} ms {v['GBps'] / 1e3:.2f} TB/s" for k, v in pk.items()))
    else:
        print(f"also {a['label']:15s}:", ffb(a))
for f in ("config2", "long_131072", "long_131072_adaptive", "long_65536", "short_4096", "short_2048", "short_1024", "short_512x4M",
          "short_512", "colshard_n524288", "colshard_n262144", "colshard_n131072", "f64_8192", "f64_long_65536"):
    print(f"{f:20s}:", ffb(line(f"r2_bench_{f}.json")))
p = line("r2_bench_panoc.json")
print("panoc standalone    :", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in p.items() if not isinstance(v, (dict, list))})
