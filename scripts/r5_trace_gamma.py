"""Per-iteration trace of ZeroFPR / PANOCplus / PANOC at BASELINE config 4's size: (iteration, gamma, tau, reads of A in the iteration, cumulative launches of
gemv_n / gemv_t / the sweeps).  What it showed (round 5): gamma goes 2.10 -> 0.0657 (five halvings) inside the FIRST iteration -- the products of the start-up --
and stays; ZeroFPR then reads A twice per iteration whatever tau its search accepts.    python scripts/r5_trace_gamma.py"""
import sys, json
sys.path.insert(0, ".")
import numpy as np
import proximalalgorithms.jl_amd as pa
m, n, dtype = 16384, 1_000_000, np.float32
ctx = pa.get_context()
A = pa.HIPMatrix.synthetic(m, n, dtype, seed=0)
rng = np.random.default_rng(12345)
k = max(1, n // 1000)
x_true = np.zeros(n, dtype); x_true[rng.choice(n, size=k, replace=False)] = rng.standard_normal(k).astype(dtype)
b = A.mul(pa.HIPVector.from_numpy(x_true)); b.axpby_(1.0, b, 0.01, pa.HIPVector.from_numpy(rng.standard_normal(m).astype(dtype)))
f = pa.LogisticLoss(b)
_, g0 = f.value_and_gradient(pa.HIPVector.zeros(m, dtype))
lam = dtype(0.1) * A.mul_adjoint(g0).norm_inf()
for name in ("ZeroFPRIteration", "PANOCplusIteration", "PANOCIteration"):
    it = getattr(pa, name)(f=f, A=A, g=pa.NormL1(lam), x0=np.zeros(n, dtype))
    rows, prev = [], 0
    ctx.profile(True); ctx.profile_reset()
    for i, s in enumerate(it):
        p = it.counters["A_passes"]; pr = ctx.profile_read()
        rows.append((i, float(s.gamma), float(s.tau), p - prev, pr["gemv_n_partial"][0], pr["gemv_t"][0], pr["gemv_tn"][0])); prev = p
        if i >= 23: break
    print(name); [print("  ", r) for r in rows]
