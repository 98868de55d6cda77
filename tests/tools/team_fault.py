#!/usr/bin/env python3
"""The long-column sweep losing a team member in the middle of a solve.

pg_ctx_test_team_fault(ctx, k, kind) makes the k-th team launch go out with one workgroup
missing: its team-mates give up after their bounded wait, the step's scalar read-back reports PG_ERR_TIMEOUT and the
iteration redoes that step with two sweeps (PG_FLAG_SWEEP_FALLBACK), then goes back to one sweep per iteration.  This
script runs FastForwardBackward (fixed or adaptive step, unsharded or as the single rank of a column-sharded job) on a
65536 x 8192 LASSO with the fault armed, next to the CPU restatement on the same inputs, and prints one JSON document:
per step the flags, the reads of A, gamma and the distance to the oracle's iterate.  Run it as its own process."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=65536)
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--mode", choices=["fixed", "adaptive"], default="fixed")
    ap.add_argument("--cols", action="store_true", help="run as the single rank of a column-sharded job (gloo, world size 1)")
    ap.add_argument("--fault", type=int, default=3, help="which team launch loses a member (0: none)")
    ap.add_argument("--refuse", action="store_true", help="the faulty launch is REFUSED (PG_ERR_UNSUPPORTED) instead of losing a member")
    ap.add_argument("--steps", type=int, default=7)
    ap.add_argument("--batched", action="store_true",
                    help="the algorithm object with device_loop=True, check_every=4: the fault lands inside a batch, which cannot be "
                         "redone -- the solve restarts with the per-iteration loop (a warning says so)")
    args = ap.parse_args()
    import proximalalgorithms.jl_amd as pa
    from oracle import proxgrad_oracle as o
    from proximalalgorithms.jl_amd import _lib

    if args.fault:
        _lib.call("pg_ctx_test_team_fault", pa.get_context().handle, args.fault, 1 if args.refuse else 0)

    m, n, dtype = args.m, args.n, np.float32
    A, b, _ = o.synthetic_lasso(m, n, seed=3, dtype=dtype)
    lam = dtype(0.1) * dtype(np.max(np.abs(A.T @ b)))
    Lf = None
    if args.mode == "fixed":  # ||A||^2 by power iteration (+10 %)
        v = np.ones(n, dtype) / dtype(np.sqrt(n))
        for _ in range(20):
            v = A.T @ (A @ v)
            v /= np.linalg.norm(v)
        Lf = dtype(1.1) * dtype(np.linalg.norm(A @ v) ** 2)
    x0 = np.zeros(n, dtype)
    comm = None
    if args.cols:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29713")
        dist.init_process_group("gloo", rank=0, world_size=1)
        comm = pa.TorchDistributedComm(shard="cols")
    f = pa.LeastSquares(A, b, comm=comm)
    iteration = pa.FastForwardBackwardIteration(f=f, g=pa.NormL1(lam), x0=x0, Lf=Lf)
    ito = iter(o.FastForwardBackwardIteration(f=o.LeastSquares(A, b), g=o.NormL1(lam), x0=x0, Lf=Lf))
    if args.batched:
        import warnings

        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            z, k = pa.FastForwardBackward(tol=0.0, maxit=args.steps + 1, device_loop=True, check_every=4)(x0=x0, f=f, g=pa.NormL1(lam), Lf=Lf)
        for _ in range(args.steps + 1):
            so = next(ito)
        print(json.dumps({"batched": True, "k": int(k), "dz": float(np.max(np.abs(z - so.z))), "z_scale": float(max(1.0, np.max(np.abs(so.z)))),
                          "warned": any("inside a batch" in str(w.message) for w in caught)}))
        return
    it = iter(iteration)
    rows, passes = [], 0
    for k in range(args.steps + 1):
        s, so = next(it), next(ito)
        z = s.z.numpy()
        p = iteration.counters.get("a_passes", 0)
        rows.append({"k": k, "flags": int(getattr(s, "flags", 0)), "a_passes": int(p - passes), "gamma": float(s.gamma),
                     "gamma_oracle": float(so.gamma), "f_x": float(s.f_x), "f_x_oracle": float(so.f_x),
                     "dz": float(np.max(np.abs(z - so.z))), "z_scale": float(max(1.0, np.max(np.abs(so.z))))})
        passes = p
    out = {"m": m, "n": n, "mode": args.mode, "cols": bool(args.cols), "fault_at_team_launch": args.fault,
           "fallback_flag": _lib.PG_FLAG_SWEEP_FALLBACK, "sweep_fallbacks": iteration.counters.get("sweep_fallbacks", 0),
           "steps": rows}
    if args.cols:
        import torch.distributed as dist

        dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
