#!/usr/bin/env python3
"""The counter rows of profiles/r6_peer_sweep_counters.md from the per-pass summaries scripts/collect_round6_profiles.sh (part `counters`)
leaves under gpurun_out/r6/ (c_<key>_<pass>.md, written by scripts/rocpd_summary.py --sum-per-dispatch): one column per kernel,
per launch, plus the derived per-16-KiB-step figures.    python scripts/r6_counter_table.py [dir]"""
import os
import sys

D = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r6")
KEYS = [("k1", "round 6 `gemv_tnp1<8,2,2,2,2>` (poll one step ahead)"), ("k1_own_step", "the same, poll in its own step"),
        ("r5", "round 5 `gemv_tnt<8,2,1,2,2,PEER,2>`"), ("tnw", "`gemv_tnw<8,4,4>` (no exchange)")]
COUNTERS = ["avg_us", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
            "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_MISC", "SQ_WAIT_ANY",
            "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
            "FETCH_SIZE", "WRITE_SIZE"]


def rows(key):
    d = {}
    for p in ("stats", "p1", "p2", "p3", "fetch", "write"):
        try:
            txt = open(os.path.join(D, "c_%s_%s.md" % (key, p))).read()
        except OSError:
            continue
        for ln in txt.splitlines():
            if "gemv_tn" not in ln or "gemv_n" in ln:
                continue
            cells = [c.strip() for c in ln.strip().strip("|").split("|")]
            if len(cells) == 7 and p == "stats":
                d["avg_us"] = float(cells[3])
            if len(cells) == 6:
                d[cells[1]] = float(cells[3])
    return d


def main():
    data = {k: rows(k) for k, _ in KEYS}
    print("| counter (per launch, summed over the device) | " + " | ".join(t for _, t in KEYS) + " |")
    print("|---|" + "---:|" * len(KEYS))
    for c in COUNTERS:
        print("| %s | " % c + " | ".join(("%.4g" % data[k][c]) if c in data[k] else "-" for k, _ in KEYS) + " |")
    # per 16 KiB of A streamed by a wave (a step of the row-team sweeps; half a step of gemv_tnw): 2048 x 2^20 x 4 B / 16 KiB = 524288 steps
    steps = 2048 * (1 << 20) * 4 / 16384
    print()
    print("| per 16 KiB of A | " + " | ".join(t for _, t in KEYS) + " |")
    print("|---|" + "---:|" * len(KEYS))
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
        print("| %s | " % c.replace("SQ_INSTS_", "") + " | ".join(("%.1f" % (data[k][c] / steps)) if c in data[k] else "-" for k, _ in KEYS) + " |")
    tot = {k: sum(data[k].get(c, 0.0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM")) for k, _ in KEYS}
    print("| all of the above | " + " | ".join("%.0f" % (tot[k] / steps) for k, _ in KEYS) + " |")
    print("| A bytes / kernel time (TB/s) | " + " | ".join(("%.2f" % (2048 * (1 << 20) * 4 / (data[k]["avg_us"] * 1e-6) / 1e12)) if "avg_us" in data[k] else "-" for k, _ in KEYS) + " |")


if __name__ == "__main__":
    main()
