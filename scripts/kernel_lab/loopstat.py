#!/usr/bin/env python3
"""instruction histogram and s_waitcnt values of the loops of one kernel in a hipcc -S listing (how the 548 / 414 instructions per step and the
vmcnt(16..21) / vmcnt(34..51) waits of profiles/r6_peer_sweep_counters.md were read):  loopstat.py file.s mangled-substring [min-len]"""
import re, sys, collections
src, key = sys.argv[1], sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 200
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") or (l.startswith("_Z") and key in l and ": ;" in l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: labels[m.group(1)] = i
def cls(op):
    if op.startswith("v_accvgpr"): return "ACC"
    if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"): return "VMEM_RD"
    if op.startswith("global_store") or op.startswith("flat_store") or op.startswith("global_atomic") or op.startswith("buffer_store"): return "VMEM_WR"
    if op.startswith("scratch_"): return "SCRATCH"
    if op.startswith("ds_"): return "LDS"
    if op.startswith("s_waitcnt"): return "WAIT"
    if op.startswith("s_load") or op.startswith("s_memrealtime") or op.startswith("s_buffer"): return "SMEM"
    if op.startswith("s_"): return "SALU"
    if op.startswith("v_"): return "VALU"
    return "OTHER"
for i, l in enumerate(body):
    m = re.match(r"^\s+(s_cbranch\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
    if m and m.group(2) in labels and labels[m.group(2)] < i and i - labels[m.group(2)] >= minlen:
        a, b = labels[m.group(2)], i
        h = collections.Counter(); ops = collections.Counter(); waits = []
        for l2 in body[a:b + 1]:
            mm = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)(;.*)?$", l2)
            if not mm or mm.group(1).startswith("."): continue
            h[cls(mm.group(1))] += 1; ops[mm.group(1)] += 1
            if mm.group(1) == "s_waitcnt": waits.append(mm.group(2).strip())
        print(f"loop {m.group(2)} lines {a}-{b} ({b - a}):", dict(h))
        print("  top ops:", ops.most_common(28))
        print("  waits:", collections.Counter(waits).most_common(20))
