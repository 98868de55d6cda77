#!/usr/bin/env python3
"""profiles/r4_team_counters.md from the passes of scripts/collect_round4_profiles.sh counters (gpurun_out/r4/c*_{team,headline}.md)."""
import os, re, sys
O = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r4"
shapes = [("team", "131072 x 131072: `gemv_tnt<16,1,4,2,2>` (teams of 8, cooperative launch)", 131072 * 131072 * 4 + (131072 + 7 * 131072) * 4),
          ("headline", "16384 x 2^20: `gemv_tnm<16,2,4,2>` (the headline)", 16384 * (1 << 20) * 4 + (16384 + 7 * (1 << 20)) * 4)]
data = {}
for key, _, _ in shapes:
    d = {}
    for p in ("cstats", "cp1", "cp2", "cp3", "cp4"):
        path = os.path.join(O, f"{p}_{key}.md")
        if not os.path.exists(path):
            continue
        for ln in open(path):
            m = re.match(r"\| `pgtn::gemv_tn\w*_kernel<[^`]*`\s*\| (\w+) \| (\d+) \| ([0-9.e+]+) \|", ln)
            if m:
                d[m.group(1)] = float(m.group(3))
            m = re.match(r"\| `pgtn::gemv_tn\w*_kernel<[^`]*` \| (\d+) \| ([0-9.]+) \| ([0-9.]+) \|", ln)
            if m and p == "cstats":
                d["_avg_us"] = float(m.group(3))
    data[key] = d
rows = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_INST_VMEM",
        "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_SCA", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
        "SQ_ACTIVE_INST_VMEM", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_SMEM", "SQ_ACTIVE_INST_MISC",
        "SQ_INST_CYCLES_VMEM", "SQ_LEVEL_WAVES", "TCC_REQ_sum", "TCC_EA0_RDREQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"]
fmt = lambda v: "—" if v is None else (f"{v:.4g}" if v < 1e5 else f"{v:.3e}")
out = ["| | " + " | ".join(s[1] for s in shapes) + " |", "|---|" + "---:|" * len(shapes)]
out.append("| kernel duration (rocprofv3 avg, us) | " + " | ".join(fmt(data[k].get("_avg_us")) for k, _, _ in shapes) + " |")
out.append("| algorithmic bytes / duration (TB/s; fraction of 8 TB/s) | " + " | ".join(
    (f"{b / (data[k]['_avg_us'] * 1e-6) / 1e12:.2f} ({b / (data[k]['_avg_us'] * 1e-6) / 8e12:.3f})" if data[k].get("_avg_us") else "—") for k, _, b in shapes) + " |")
for r in rows:
    if any(r in data[k] for k, _, _ in shapes):
        out.append(f"| {r} | " + " | ".join(fmt(data[k].get(r)) for k, _, _ in shapes) + " |")
def share(k, num):
    d = data[k]
    return f"{d[num] / d['SQ_WAVE_CYCLES']:.2f}" if num in d and "SQ_WAVE_CYCLES" in d else "—"
out.append("| **share of wave time** parked on s_waitcnt / barrier (WAIT_ANY) | " + " | ".join(share(k, "SQ_WAIT_ANY") for k, _, _ in shapes) + " |")
out.append("| ... stalled at issue (WAIT_INST_ANY; of which LDS) | " + " | ".join(f"{share(k, 'SQ_WAIT_INST_ANY')} ({share(k, 'SQ_WAIT_INST_LDS')})" for k, _, _ in shapes) + " |")
out.append("| ... issuing (ACTIVE_INST_ANY; of which VALU / scalar / LDS) | " + " | ".join(
    f"{share(k, 'SQ_ACTIVE_INST_ANY')} ({share(k, 'SQ_ACTIVE_INST_VALU')} / {share(k, 'SQ_ACTIVE_INST_SCA')} / {share(k, 'SQ_ACTIVE_INST_LDS')})" for k, _, _ in shapes) + " |")
out.append("| TCC_EA0_RDREQ x 128 B / algorithmic bytes | " + " | ".join(
    (f"{data[k]['TCC_EA0_RDREQ_sum'] * 128 / b:.3f}" if "TCC_EA0_RDREQ_sum" in data[k] else "—") for k, _, b in shapes) + " |")
out.append("| SALU instructions per VMEM read instruction | " + " | ".join(
    (f"{data[k]['SQ_INSTS_SALU'] / data[k]['SQ_INSTS_VMEM_RD']:.2f}" if "SQ_INSTS_SALU" in data[k] and "SQ_INSTS_VMEM_RD" in data[k] else "—") for k, _, _ in shapes) + " |")
out.append("| VALU instructions per VMEM read instruction | " + " | ".join(
    (f"{data[k]['SQ_INSTS_VALU'] / data[k]['SQ_INSTS_VMEM_RD']:.2f}" if "SQ_INSTS_VALU" in data[k] and "SQ_INSTS_VMEM_RD" in data[k] else "—") for k, _, _ in shapes) + " |")
print("\n".join(out))
