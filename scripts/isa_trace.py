#!/usr/bin/env python3
"""Memory / wait skeleton of one kernel in a `hipcc -S --cuda-device-only` listing: runs of tile loads are collapsed, every
s_waitcnt, barrier, small load, store and scratch access is shown with its line.  Usage: isa_trace.py file.s <symbol substring>
(how the in-order return of vector loads, stray vmcnt(0) and AGPR copies behind a load were found in the sweep kernels)."""
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l and l.rstrip().endswith(tuple("E:")) or (l.startswith("_Z") and pat in l.split(":")[0] and ":" in l))
out, n4 = [], 0
for i in range(start, len(lines)):
    l = lines[i].strip()
    if l.startswith("s_endpgm"):
        break
    if l.startswith("global_load_dwordx4") or l.startswith("global_load_dwordx2"):
        n4 += 1
        continue
    if re.match(r"(\.LBB\S+:|s_cbranch)", l) and "-b" not in sys.argv:
        continue  # branches are transparent unless -b is given
    keep = re.match(r"(s_waitcnt|s_barrier|global_load|global_store|scratch_|ds_read|ds_write|s_load|\.LBB\S+:|s_cbranch|s_sleep)", l)
    if keep:
        if n4:
            out.append(f"        ... {n4} wide global loads")
            n4 = 0
        out.append(f"{i - start:6d}  {l[:90]}")
print("\n".join(out))
