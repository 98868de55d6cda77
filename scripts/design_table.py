#!/usr/bin/env python3
"""Write the "Measured" table of DESIGN.md from the bench lines under profiles/ (run after
scripts/publish_round6_profiles.sh): the table is transcribed by a program, not by hand.  Every line is taken from the NEWEST
round that collected it (profiles/r6_<name>, else r5_, r4_, r3_); the source column names the file actually used, and the
rounds used are printed in the table's first line.
    python scripts/design_table.py           print the table
    python scripts/design_table.py --apply   replace the block between the <!-- measured:begin/end --> markers of DESIGN.md"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


USED = {}


def pick(name):
    """the newest round that collected <name>"""
    for rnd in ("r6", "r5", "r4", "r3"):
        if os.path.exists(os.path.join(P, f"{rnd}_{name}")):
            USED[name] = rnd
            return f"{rnd}_{name}"
    raise SystemExit(name + ": not collected")


def line(name):
    name = pick(name[3:]) if name.startswith("r3_") else name
    for l in open(os.path.join(P, name)):
        if l.startswith("{"):
            return json.loads(l)
    raise SystemExit(name + ": no JSON line")


def src(*names):
    """`rX_name` for the source column (after the lines were read)"""
    return ", ".join("`%s_%s`" % (USED.get(n, "r3"), n) for n in names)


def tb(r):
    return r["achieved"] / 1e3


def kern(r):
    return r["roofline"].get("kernel_name") or r["roofline"]["kernel"]


def main():
    d = line("r3_bench_default.json")
    also = {a["label"]: a for a in d["also"]}
    r = d["roofline"]
    ceiling = None
    for l in open(os.path.join(P, pick("stream_ceiling.log"))):
        m = re.search(r"wave-contiguous runs \(the sweeps' pattern\) ([0-9.]+) GB/s", l)
        if m:
            ceiling = float(m.group(1)) / 1e3
    rows = []
    add = lambda *c: rows.append("| " + " | ".join(c) + " |")
    add("workload (Float32 unless noted)", "it/s", "sweep kernel: avg launch (HIP events)", "A bytes ÷ kernel time", "of 8 TB/s", "source (`profiles/`)")
    add("---", "---", "---", "---", "---", "---")
    sus = d.get("sustained")
    add("headline 16384 × 2^20, fixed step (the driver's line)",
        f"**{d['value']:.1f}** ({d['ms_per_step']:.2f} ms/step" + (f"; kept running for {sus['seconds']:.0f} s: {sus['value']:.1f}" if sus else "") + ")",
        f"`gemv_tnm<16,2,4>` {r['avg_launch_ms']:.2f} ms", f"{tb(r):.2f} TB/s", f"**{r['frac']:.3f}**", src("bench_default.json"))
    a = also["headline_adaptive"]
    add("the same, adaptive step (`benchmarks.jl:55-61`)", f"{a['value']:.1f}", f"{a['roofline']['avg_launch_ms']:.2f} ms", f"{tb(a['roofline']):.2f} TB/s",
        f"{a['roofline']['frac']:.3f}", "`also[0]` of the same line")
    c2, a2 = line("r3_bench_config2.json"), also["config2"]
    add("config 2, 8192 × 262144", f"{c2['value']:.0f} ({a2['value']:.0f} inside `also`)", f"`gemv_tnm<4,4,8>` {c2['roofline']['avg_launch_ms']:.3f} ms",
        f"{tb(c2['roofline']):.2f} TB/s", f"{c2['roofline']['frac']:.3f} ({a2['roofline']['frac']:.3f})", src("bench_config2.json") + ", `also[1]`")
    c3 = also["config3"]
    s3, l3 = c3["stepping"], c3["device_loop"]
    r3 = s3["roofline"]
    b2b = r3.get("back_to_back_ms")
    pair_ms, pair_frac = r3.get("event_pair_ms", r3["avg_launch_ms"]), r3.get("event_pair_frac", r3["frac"])  # (older lines: avg_launch_ms was the pair)
    add("config 3, DouglasRachford n = 10^7: stepping", f"{s3['value'] / 1e3:.1f} k",
        f"`dr_step` " + (f"{b2b * 1e3:.1f} µs back to back (= rocprof), " if b2b else "") + f"{pair_ms * 1e3:.1f} µs with an event pair per launch (200 MB)",
        (f"{0.2 / b2b:.2f} / " if b2b else "") + f"{0.2 / pair_ms:.2f} TB/s", (f"**{r3['back_to_back_frac']:.3f}** / " if b2b else "") + f"{pair_frac:.3f}",
        "`also[2]`, `r3_dr_counters.md`")
    kk = l3["iterations_per_launch"]
    add(f"config 3: in-library loop, {kk} iterations per sweep, two sweeps in flight", f"**{l3['value'] / 1e3:.1f} k**",
        f"`dr_block<{kk}>` {l3['roofline']['avg_launch_ms'] * 1e3:.0f} µs",
        f"VALU-bound: {l3['valu']['frac']:.2f} of the loop's VALU issue floor ({l3['valu']['floor_ms_per_launch'] * 1e3:.0f} µs)", "—", "`also[2].device_loop`")
    c4 = also["config4"]
    pick("bench_panoc.json")
    pk = c4["roofline"]["per_kernel"]
    ks = [k for k in ("gemv_tn", "gemv_n_partial") if k in pk]
    add("config 4, PANOC logistic + L1 16384 × 10^6, L-BFGS(5), adaptive (round 3: 54.9 it/s at 2.0 reads)",
        f"**{c4['value']:.1f}** ({c4['config']['A_passes_per_step']:.1f} reads of A per iteration)",
        ", ".join(("sweep" if k == "gemv_tn" else f"`{k}`") + f" {pk[k]['avg_ms']:.2f} ms" for k in ks),
        " / ".join(f"{pk[k]['GBps'] / 1e3:.2f}" for k in ks) + " TB/s", " / ".join(f"{pk[k]['GBps'] / 8e3:.3f}" for k in ks),
        "`also[3]`, " + src("bench_panoc.json"))
    try:  # round 5: ZeroFPR with three / two trial points of its line search per sweep, beside one per sweep (round 4), and PANOCplus
        zf, z2, z1 = line("r3_bench_zerofpr.json"), line("r3_bench_zerofpr_two_points.json"), line("r3_bench_zerofpr_single_trials.json")
        pp, p2 = line("r3_bench_panocplus.json"), line("r3_bench_panocplus_two_reads.json")
        add("config 4's family, first 23 iterations: ZeroFPR with three / two trial points per sweep / one per sweep (round 4); PANOCplus with its second pass in the next first sweep / as a pass of its own (round 4)",
            f"**{zf['value']:.1f}** / {z2['value']:.1f} / {z1['value']:.1f}; **{pp['value']:.1f}** / {p2['value']:.1f}",
            f"reads of A per iteration {zf['A_passes_per_step']:.2f} / {z2['A_passes_per_step']:.2f} / {z1['A_passes_per_step']:.2f}; {pp['A_passes_per_step']:.2f} / {p2['A_passes_per_step']:.2f}",
            " / ".join(f"{x['roofline']['achieved'] / 1e3:.2f}" for x in (zf, z2, z1, pp, p2)) + " TB/s (all sweeps)",
            " / ".join(f"{x['roofline']['frac']:.3f}" for x in (zf, z2, z1, pp, p2)),
            src("bench_zerofpr.json", "bench_zerofpr_two_points.json", "bench_zerofpr_single_trials.json", "bench_panocplus.json", "bench_panocplus_two_reads.json"))
    except SystemExit:
        pass
    try:  # round 5: north_star's row layout as `bench.py --gpus 2` reports it (two rank PROCESSES on this one device, gloo)
        for key, nrows in (("2rank_rows_2048", 2048), ("2rank_rows_16384", 16384)):
            t = line(f"r3_bench_{key}.json")
            two = t.get("rows_two_sweeps") or {}
            cfg = t["config"]
            agg = t["value"] * cfg["m"] * cfg["n"] * 4 / 1e12
            add(f"row layout, 2 processes x {nrows} rows on this device: top-level record (`row_layout` = {cfg.get('row_layout')}) / the two-sweep record it replaced",
                f"**{t['value']:.1f}** / {two.get('value', float('nan')):.1f}", f"sweep {t['roofline']['avg_launch_ms']:.2f} ms per rank",
                f"{agg:.2f} TB/s (A per iteration, both ranks)", f"{agg / 8:.3f}", src(f"bench_{key}.json") + ", `r6_row_team_latency_sweep.md`")
    except (SystemExit, KeyError):
        pass
    lf, la, l6 = line("r3_bench_long_131072.json"), line("r3_bench_long_131072_adaptive.json"), line("r3_bench_long_65536.json")
    add("long columns 131072 × 131072 (config 5's per-GPU block under column shards), fixed / adaptive",
        f"**{lf['value']:.1f} / {la['value']:.1f}** with ONE read of A", f"`gemv_tnt<16,1,4,2,2>` (cooperative launch) {lf['roofline']['avg_launch_ms']:.2f} ms",
        f"{tb(lf['roofline']):.2f} TB/s", f"**{lf['roofline']['frac']:.3f}**", src("bench_long_131072.json", "bench_long_131072_adaptive.json") + ", `also[4]`")
    add("long columns 65536 × 262144", f"{l6['value']:.1f}", f"{l6['roofline']['avg_launch_ms']:.2f} ms", f"{tb(l6['roofline']):.2f} TB/s",
        f"{l6['roofline']['frac']:.3f}", src("bench_long_65536.json"))
    od = [line(f"r3_bench_odd_{k}.json") for k in ("50000", "100000", "10000")]
    add("column lengths that fill no power of two: 50000 × 84000 / 100000 × 84000 (exact-`U` team members) / 10000 × 420000 (`gemv_tnm<10,4,4>`)",
        " / ".join(f"{x['value']:.0f}" for x in od), " / ".join(f"{x['roofline']['avg_launch_ms']:.2f}" for x in od) + " ms",
        " / ".join(f"{tb(x['roofline']):.2f}" for x in od) + " TB/s", " / ".join(f"{x['roofline']['frac']:.3f}" for x in od) + " (round 3: 0.824 / 0.866 / 0.868)",
        src("bench_odd_50000.json", "bench_odd_100000.json", "bench_odd_10000.json"))
    md = [line(f"r3_bench_mid_{k}.json") for k in ("7168", "10240", "12288", "24576", "32768")]
    add("mid-length columns 7168 × 299593 (`gemv_tn<4,8,8>`) / 10240 × 209715 / 12288 × 174762 / 24576 × 87381 / 32768 × 65536 (`gemv_tnm`, `U` = 10 / 12 / 12 / 16)",
        " / ".join(f"{x['value']:.0f}" for x in md), " / ".join(f"{x['roofline']['avg_launch_ms']:.3f}" for x in md) + " ms",
        " / ".join(f"{tb(x['roofline']):.2f}" for x in md) + " TB/s", "**" + " / ".join(f"{x['roofline']['frac']:.3f}" for x in md) + "**", src("bench_mid_7168.json") + " ...")
    sh = [line(f"r3_bench_short_{k}.json") for k in ("4096", "2048", "1024", "512x4M", "512")]
    add("short columns 4096 (`gemv_tnc`) / 2048 / 1024 × 2^20 / 512 × 2^22 / 512 × 2^20 (`gemv_tnw`; row-shard shapes of N = 8 and below)",
        " / ".join(f"{x['value']:.0f}" for x in sh), " / ".join(f"{x['roofline']['avg_launch_ms']:.3f}" for x in sh) + " ms",
        " / ".join(f"{tb(x['roofline']):.2f}" for x in sh) + " TB/s", " / ".join(f"{x['roofline']['frac']:.3f}" for x in sh), src("bench_short_4096.json", "bench_short_2048.json", "bench_short_1024.json") + " ..., `also[5]`")
    cs = [line(f"r3_bench_colshard_n{k}.json") for k in ("524288", "262144", "131072")]
    add("column-shard shapes with the collective attached (one rank, native RCCL communicator): n = 2^19 / 2^18 / 2^17",
        " / ".join(f"{x['value']:.1f}" for x in cs) + " (" + " / ".join(f"{x['value'] / d['value']:.2f}×" for x in cs) +
        " this box's one-GPU rate before the 64 KiB all-reduce)",
        " / ".join(f"{x['roofline']['avg_launch_ms']:.2f}" for x in cs) + " ms", "",
        " / ".join(f"{x['roofline']['frac']:.3f}" for x in cs), src("bench_colshard_n524288.json") + " ..., `r3_colshard_step_trace*.md`")
    f8, fl = line("r3_bench_f64_8192.json"), line("r3_bench_f64_long_65536.json")
    add("Float64: 8192 × 2^20 (`gemv_tnm<double,16,2,4>`) / long columns 65536 × 131072", f"{f8['value']:.1f} / {fl['value']:.1f}",
        f"{f8['roofline']['avg_launch_ms']:.2f} / {fl['roofline']['avg_launch_ms']:.2f} ms", f"{tb(f8['roofline']):.2f} / {tb(fl['roofline']):.2f} TB/s",
        f"{f8['roofline']['frac']:.3f} / {fl['roofline']['frac']:.3f}", src("bench_f64_8192.json", "bench_f64_long_65536.json"))
    cb = d["cpu_baseline"]
    npb = cb.get("numpy_openblas", {})
    tw = cb.get("c_openmp_twin") or {"value": cb["value"], "cores": cb["cores"]}  # present when OpenBLAS was the faster one (= `value`)
    twin_gbps = 2.0 * 16384 * (1 << 20) * 4 * tw["value"] / 1e9
    add(f"CPU: the C / OpenMP twin on the same 64 GiB matrix, {tw['cores']} threads / one thread; numpy + OpenBLAS oracle "
        f"(`cpu_baseline.value` = the faster of the two: {cb['value']:.2f})",
        f"{tw['value']:.2f} / {cb['value_1thread']:.2f}; {npb.get('value', float('nan')):.2f}", "", f"{twin_gbps:.0f} / {cb.get('achieved_GBps_1thread', 0):.0f} GB/s; {npb.get('achieved_GBps', 0):.0f} GB/s",
        (f"host read pass over the same matrix {cb['host_read_GBps']} GB/s; " if cb.get("host_read_GBps") else "") + f"STREAM add (numpy) {cb.get('host_stream_GBps')} GB/s",
        "`cpu_baseline` of the line")
    head = (f"streaming-read ceiling of the same box ({src('stream_ceiling.log')}, random data): {ceiling:.2f} TB/s — the headline sweep is at "
            f"{tb(r) / ceiling:.2f} of it, config 2 {tb(c2['roofline']) / ceiling:.2f}, long columns {tb(lf['roofline']) / ceiling:.2f}, "
            f"2048-row columns {tb(sh[1]['roofline']) / ceiling:.2f}" if ceiling else "")
    text = head + "\n\n" + "\n".join(rows) + "\n"
    if "--apply" in sys.argv:
        path = os.path.join(ROOT, "DESIGN.md")
        s = open(path).read()
        b, e = "<!-- measured:begin -->\n", "<!-- measured:end -->\n"
        i, j = s.index(b) + len(b), s.index(e)
        open(path, "w").write(s[:i] + text + s[j:])
    else:
        print(text)


if __name__ == "__main__":
    main()
