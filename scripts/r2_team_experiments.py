"""Timing experiments on the long-column sweep (library built with PG_EXTRA_HIPCC_FLAGS=-DPG_TNT_EXPERIMENT; results of
the perturbed runs are WRONG by construction): where does the gap to the one-workgroup kernel go?
  PG_TNT_DBG bit 0: never wait for the other members; bit 1: accumulate from the register tile (no LDS read-back);
  bit 2: do not park tiles in LDS."""
import os, sys

os.environ.setdefault("PG_TUNE", "1")  # the library reads its tuning variables only when this is set

import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proximalalgorithms.jl_amd as pa
from scripts.r2_tn_check import setup, time_pass, clear
ctx = pa.get_context()
g = pa.NormL1(0.3)
for (m, n) in ((131072, 131072), (32768, 524288)):
    A, f, x, vs = setup(m, n)
    nbytes = m * n * 4
    print(f"=== {m}x{n} ===")
    for dbg in (0, 1, 2, 4, 6, 7):
        clear()
        os.environ["PG_TNT_DBG"] = str(dbg)
        if m * 4 <= 128 * 1024:
            os.environ["PG_TN_KERNEL"] = "team"
        ms = time_pass(f, x, vs, g, ctx, reps=6)
        print(f"  dbg={dbg}: {ms:.3f} ms {nbytes / (ms * 1e-3) / 1e9:.0f} GB/s")
    os.environ.pop("PG_TNT_DBG", None); os.environ.pop("PG_TN_KERNEL", None)
    if m * 4 <= 128 * 1024:
        ms = time_pass(f, x, vs, g, ctx, reps=6)
        print(f"  one-workgroup kernel: {ms:.3f} ms {nbytes / (ms * 1e-3) / 1e9:.0f} GB/s")
    del f, A
