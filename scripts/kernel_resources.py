#!/usr/bin/env python3
"""Registers / scratch / occupancy of every kernel in one HIP source (hipcc -Rpass-analysis=kernel-resource-usage):
    python scripts/kernel_resources.py proximalalgorithms.jl_amd/csrc/pg_gemv_tn2.hip [name-filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
       "-I", os.path.join(ROOT, "proximalalgorithms.jl_amd", "csrc"), "-fno-gpu-rdc", "-Rpass-analysis=kernel-resource-usage",
       "-c", src, "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for ln in out.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?)\s+\[-Rpass", ln) or re.search(r":\d+:\d+: remark:\s+(.*?)\s+\[-Rpass", ln)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", "")).replace("void ", "")
    if flt and flt not in name:
        continue
    print(f"{name:70s} vgpr {r.get('VGPRs','?'):>4s} agpr {r.get('AGPRs','?'):>3s} sgpr {r.get('SGPRs','?'):>4s} "
          f"scratch {r.get('ScratchSize [bytes/lane]','?'):>5s} occ {r.get('Occupancy [waves/SIMD]','?'):>2s} lds {r.get('LDS Size [bytes/block]','?')}")
