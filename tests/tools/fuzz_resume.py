#!/usr/bin/env python3
"""Randomised check of checkpoint / resume (pg_iter_state_download / _upload): on a random LASSO-type problem
with random iteration options, k1 iterations + save + a NEW iterator that takes the blob + k2 iterations must equal k1 + k2
iterations straight, BIT FOR BIT in every state vector and scalar -- the reference's `iterate(iter, saved_state)` continues
from any state because all algorithm memory is in the state struct (fast_forward_backward.jl:60-71, nesterov.jl:56-60).
Usage: python tests/tools/fuzz_resume.py [cases] [first_seed].  Column lengths are drawn so that every geometry of the
single sweep holds a speculative half iteration at the save point (one wave per column group, shared workgroups, teams)."""
import gc
import itertools
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import proximalalgorithms.jl_amd as pa  # noqa: E402


def one_case(seed):
    rng = np.random.default_rng(seed)
    dtype = np.float32 if rng.random() < 0.6 else np.float64
    if rng.random() < 0.5:
        m, n = int(rng.integers(1, 700)), int(rng.integers(1, 900))
    else:  # every sweep geometry: <= 8 row groups, 9..28, 29..128, teams
        m = int(rng.choice([300, 2048, 2305, 4096, 7000, 8192, 16384, 20000, 33000, 40000, 70001]))
        n = int(rng.choice([1, 2, 7, 33, 64, 130]))
    fast = bool(rng.random() < 0.65)
    adaptive = bool(rng.random() < 0.5)
    gname = rng.choice(["l1", "l1w", "box", "boxv", "zero"], p=[0.4, 0.15, 0.15, 0.15, 0.15])
    A = np.asfortranarray(rng.standard_normal((m, n)).astype(dtype) / dtype(np.sqrt(m)))
    xt = np.zeros(n, dtype)
    nzc = max(1, n // 10)
    xt[rng.choice(n, nzc, replace=False)] = rng.standard_normal(nzc).astype(dtype)
    b = (A @ xt + dtype(0.01) * rng.standard_normal(m).astype(dtype)).astype(dtype)
    lam = dtype(0.1) * dtype(max(np.max(np.abs(A.T @ b)), 1e-3))
    Lf = None if adaptive else dtype(1.02 * np.linalg.norm(A.astype(np.float64), 2) ** 2)
    x0 = (0.1 * rng.standard_normal(n)).astype(dtype)
    lo = (-0.3 - 0.2 * rng.random(n)).astype(dtype)
    hi = (lo + dtype(0.7)).astype(dtype)
    w = (lam * (0.25 + 1.5 * rng.random(n))).astype(dtype)
    make_g = {"l1": lambda: pa.NormL1(lam), "l1w": lambda: pa.NormL1(w), "box": lambda: pa.IndBox(dtype(-0.3), dtype(0.4)),
              "boxv": lambda: pa.IndBox(lo, hi), "zero": lambda: pa.Zero()}[gname]
    kw = {}
    seq = "default"
    if fast:
        seq = str(rng.choice(["default", "fixed", "simple", "constant", "repeated", "host"]))
        mf = dtype(0.05) if seq == "constant" or rng.random() < 0.2 else dtype(0)
        kw["mf"] = mf
        if seq == "fixed":
            kw["extrapolation_sequence"] = lambda: pa.FixedNesterovSequence(dtype)
        elif seq == "simple":
            kw["extrapolation_sequence"] = lambda: pa.SimpleNesterovSequence(dtype)
        elif seq == "constant":
            kw["extrapolation_sequence"] = lambda: pa.ConstantNesterovSequence(dtype(0.05), dtype(0.5))
        elif seq == "repeated":
            kw["extrapolation_sequence"] = lambda: itertools.repeat(dtype(0.25))
    if adaptive:
        kw["increase_gamma"] = dtype(rng.choice([1.0, 1.01]))
        if fast:
            kw["reuse_residual"] = bool(rng.random() < 0.75)
    kw["single_sweep"] = bool(rng.random() < 0.75)
    k1, k2 = int(rng.integers(1, 25)), int(rng.integers(1, 12))
    label = (f"seed={seed} {np.dtype(dtype).name} {m}x{n} fast={fast} adaptive={adaptive} g={gname} seq={seq} "
             f"{ {k: v for k, v in kw.items() if k != 'extrapolation_sequence'} } k1={k1} k2={k2}")
    if seq == "host":  # a host-fed sequence lives in the CALLER's iterator: its state is not part of the blob (documented)
        return None, label
    cls = pa.FastForwardBackwardIteration if fast else pa.ForwardBackwardIteration
    f = pa.LeastSquares(A, b)

    def make():
        k = dict(kw)
        if "extrapolation_sequence" in k:
            k["extrapolation_sequence"] = k["extrapolation_sequence"]()
        if not fast:
            k.pop("mf", None)
        return cls(f=f, g=make_g(), x0=x0, Lf=Lf, engine="fused", **k)  # (the automatic choice is another engine for a large adaptive FB)

    fields = ("x", "grad_f_x", "y", "z", "res") + (("z_prev",) if fast else ())

    def snap(s):
        d = {k: getattr(s, k).numpy().copy() for k in fields}
        d.update(gamma=float(s.gamma), f_x=float(s.f_x), g_z=float(s.g_z), res_inf=float(s.res_inf))
        return d

    straight = [snap(s) for s in itertools.islice(make(), k1 + k2)]
    first = make()
    it = iter(first)
    for _ in range(k1):
        s = next(it)
    blob = first.save_state()
    del it, s, first
    gc.collect()
    resumed = make()
    for k, s in enumerate(itertools.islice(resumed.resume(blob), k2), start=k1):
        got, ref = snap(s), straight[k]
        for name in ref:
            same = np.array_equal(got[name], ref[name], equal_nan=True) if isinstance(ref[name], np.ndarray) else (
                got[name] == ref[name] or (np.isnan(got[name]) and np.isnan(ref[name])))
            if not same:
                return f"{name} differs at iteration {k}", label
    return "", label


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    pa.get_context()
    t0 = time.time()
    bad = skipped = 0
    for seed in range(seed0, seed0 + cases):
        try:
            why, label = one_case(seed)
        except Exception as e:  # noqa: BLE001 -- a crash is a finding
            why, label = f"{type(e).__name__}: {e}", f"seed={seed}"
        if why is None:
            skipped += 1
        elif why:
            bad += 1
            print("FAIL", label, "--", why, flush=True)
    print(f"{cases} cases ({skipped} drawn with a host-fed sequence and skipped), {bad} failing, {time.time() - t0:.1f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
