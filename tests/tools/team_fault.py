#!/usr/bin/env python3
"""The long-column sweep losing a team member in the middle of a solve.

pg_ctx_test_team_fault(ctx, k, kind) makes the k-th team launch go out with one workgroup
missing: its team-mates give up after their bounded wait, the step's scalar read-back reports PG_ERR_TIMEOUT and the
iteration redoes that step with two sweeps (PG_FLAG_SWEEP_FALLBACK), then goes back to one sweep per iteration.  This
script runs FastForwardBackward (fixed or adaptive step, unsharded or as the single rank of a column-sharded job, stepped or
batched, the fault a lost member or a refused launch: --cases) on a 65536 x n LASSO with the fault armed, next to the CPU
restatement on the same inputs, and prints one JSON document {case: per step the flags, the reads of A, gamma and the
distance to the oracle's iterate}.  Run it as its own process (all cases share the matrix and the oracle runs)."""
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
    ap.add_argument("--fault", type=int, default=3, help="which team launch loses a member (0: none)")
    ap.add_argument("--steps", type=int, default=7)
    ap.add_argument("--cases", default="fixed",
                    help="comma-separated cases run in this ONE process on the same matrix (generated once, one oracle run per step "
                         "mode): <mode>[-cols][-batched][-refuse] with mode fixed | adaptive; cols = the single rank of a "
                         "column-sharded job (gloo, world size 1); batched = the algorithm object with device_loop=True, "
                         "check_every=4 (the fault lands inside a batch); refuse = the faulty launch is REFUSED "
                         "(PG_ERR_UNSUPPORTED) instead of losing a member")
    args = ap.parse_args()
    import itertools
    import warnings

    import proximalalgorithms.jl_amd as pa
    from oracle import proxgrad_oracle as o
    from proximalalgorithms.jl_amd import _lib

    m, n, dtype = args.m, args.n, np.float32
    A, b, _ = o.synthetic_lasso(m, n, seed=3, dtype=dtype)
    lam = dtype(0.1) * dtype(np.max(np.abs(A.T @ b)))
    v = np.ones(n, dtype) / dtype(np.sqrt(n))  # ||A||^2 by power iteration (+10 %)
    for _ in range(20):
        v = A.T @ (A @ v)
        v /= np.linalg.norm(v)
    Lf_fixed = dtype(1.1) * dtype(np.linalg.norm(A @ v) ** 2)
    x0 = np.zeros(n, dtype)
    ctx = pa.get_context()
    A_dev = pa.HIPMatrix.from_numpy(A, ctx)
    b_dev = pa.HIPVector.from_numpy(b, ctx)
    cases = [c.split("-") for c in args.cases.split(",")]
    cases.sort(key=lambda c: "cols" in c)  # the collective is attached to the context once, for the column-sharded cases at the end
    oracle_states = {}

    def oracle(mode):
        if mode not in oracle_states:
            Lf = Lf_fixed if mode == "fixed" else None
            it = o.FastForwardBackwardIteration(f=o.LeastSquares(A, b), g=o.NormL1(lam), x0=x0, Lf=Lf)
            oracle_states[mode] = [(s.z.copy(), float(s.gamma), float(s.f_x)) for s in itertools.islice(it, args.steps + 1)]
        return oracle_states[mode]

    comm = None
    out = {}
    for case in cases:
        mode, cols, batched, refuse = case[0], "cols" in case, "batched" in case, "refuse" in case
        Lf = Lf_fixed if mode == "fixed" else None
        ref = oracle(mode)
        if cols and comm is None:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29713")
            dist.init_process_group("gloo", rank=0, world_size=1)
            comm = pa.TorchDistributedComm(shard="cols")
        f = pa.LeastSquares(A_dev, b_dev, comm=comm if cols else None)
        _lib.call("pg_ctx_test_team_fault", ctx.handle, args.fault, 1 if refuse else 0)  # counts team launches from here
        key = "-".join(case)
        if batched:
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                z, k = pa.FastForwardBackward(tol=0.0, maxit=args.steps + 1, device_loop=True, check_every=4)(x0=x0, f=f, g=pa.NormL1(lam), Lf=Lf)
            zo = ref[args.steps][0]
            out[key] = {"batched": True, "k": int(k), "dz": float(np.max(np.abs(z - zo))), "z_scale": float(max(1.0, np.max(np.abs(zo)))),
                        "warned": any("inside a batch" in str(w.message) for w in caught)}
            continue
        iteration = pa.FastForwardBackwardIteration(f=f, g=pa.NormL1(lam), x0=x0, Lf=Lf)
        rows, passes = [], 0
        for k, s in enumerate(itertools.islice(iteration, args.steps + 1)):
            z = s.z.numpy()
            zo, go, fo = ref[k]
            p = iteration.counters.get("a_passes", 0)
            rows.append({"k": k, "flags": int(getattr(s, "flags", 0)), "a_passes": int(p - passes), "gamma": float(s.gamma),
                         "gamma_oracle": go, "f_x": float(s.f_x), "f_x_oracle": fo,
                         "dz": float(np.max(np.abs(z - zo))), "z_scale": float(max(1.0, np.max(np.abs(zo))))})
            passes = p
        out[key] = {"m": m, "n": n, "mode": mode, "cols": cols, "fault_at_team_launch": args.fault,
                    "fallback_flag": _lib.PG_FLAG_SWEEP_FALLBACK, "sweep_fallbacks": iteration.counters.get("sweep_fallbacks", 0),
                    "steps": rows}
        del iteration
    _lib.call("pg_ctx_test_team_fault", ctx.handle, 0, 0)
    if comm is not None:
        import torch.distributed as dist

        dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
