"""Long-column sweep: prefetch distance (PG_TNT_PF) x member shape (PG_TNT_U / PG_TNT_C)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proximalalgorithms.jl_amd as pa
from scripts.r2_tn_check import setup, time_pass, clear, check_one
import numpy as np
ok = True
for env in ({"PG_TNT_PF": "2", "PG_TNT_U": "8", "PG_TNT_C": "1"}, {"PG_TNT_PF": "2", "PG_TNT_U": "4", "PG_TNT_C": "2"}, {"PG_TNT_PF": "2", "PG_TNT_LAG": "1", "PG_TNT_U": "8", "PG_TNT_C": "1"},
            {"PG_TNT_U": "8", "PG_TNT_C": "2", "PG_TNT_LAG": "1"}, {"PG_TNT_U": "8", "PG_TNT_C": "2", "PG_TNT_LAG": "0"}, {}):
    for (m, n) in ((256 * 256, 333), (300 * 256 + 17, 700), (512 * 256, 24)):
        ok &= check_one(m, n, np.float32, env)
    ok &= check_one(256 * 128, 333, np.float64, env)
print("PF variants correct" if ok else "PF VARIANTS WRONG")
ctx = pa.get_context()
g = pa.NormL1(0.3)
for (m, n) in ((131072, 131072), (65536, 262144)):
    A, f, x, vs = setup(m, n)
    nbytes = m * n * 4
    print(f"=== {m}x{n} ===")
    for env in ({}, {"PG_TNT_U": "8", "PG_TNT_C": "1"}, {"PG_TNT_PF": "2", "PG_TNT_U": "8", "PG_TNT_C": "1"},
                {"PG_TNT_U": "4", "PG_TNT_C": "2"}, {"PG_TNT_PF": "2", "PG_TNT_U": "4", "PG_TNT_C": "2"},
                {"PG_TNT_U": "8", "PG_TNT_C": "2", "PG_TNT_LAG": "1"}, {"PG_TNT_U": "8", "PG_TNT_C": "2", "PG_TNT_LAG": "0"},
                {"PG_TNT_PF": "2", "PG_TNT_LAG": "1", "PG_TNT_U": "8", "PG_TNT_C": "1"}, {"PG_TNT_U": "8", "PG_TNT_C": "1", "PG_TNT_LAG": "1"}):
        clear(); os.environ.pop("PG_TNT_PF", None)
        os.environ.update(env)
        try:
            ms = time_pass(f, x, vs, g, ctx, reps=6)
            print(f"  {str(env):64s} {ms:.3f} ms {nbytes / (ms * 1e-3) / 1e9:.0f} GB/s")
        except pa.ProxGradError as e:
            print(f"  {str(env):64s} error {str(e)[:60]}")
    os.environ.pop("PG_TNT_PF", None)
    del f, A
