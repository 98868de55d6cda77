#!/usr/bin/env python3
"""Float32 accuracy of the two GEMV passes at the headline size, measured against the Float64 kernels on the same
(bit-identical, 128 GiB in f64) matrix: relative error of r = A x - b entries, of f = ||r||^2 / 2 and of g = A' r."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proximalalgorithms.jl_amd as pa

m, n = 16384, 1 << 20
rng = np.random.default_rng(0)
x = np.zeros(n, np.float32); idx = rng.choice(n, n // 50, replace=False); x[idx] = rng.standard_normal(idx.size).astype(np.float32)
xd = rng.standard_normal(n).astype(np.float32)   # a dense x as the worst case
b = rng.standard_normal(m).astype(np.float32)
out = {}
res = {}
for dt in (np.float32, np.float64):
    A = pa.HIPMatrix.synthetic(m, n, dt, seed=0)
    f = pa.LeastSquares(A, pa.HIPVector.from_numpy(b.astype(dt)))
    for name, xv in (("sparse_x", x), ("dense_x", xd)):
        fx, g = f.value_and_gradient(pa.HIPVector.from_numpy(xv.astype(dt)))
        res[(np.dtype(dt).name, name)] = (float(fx), f.residual().numpy().astype(np.float64), g.numpy().astype(np.float64))
    del f, A
for name in ("sparse_x", "dense_x"):
    f32, r32, g32 = res[("float32", name)]; f64, r64, g64 = res[("float64", name)]
    out[name] = {"rel_err_f": abs(f32 - f64) / abs(f64), "max_abs_err_r_over_rms_r": float(np.max(np.abs(r32 - r64)) / np.sqrt(np.mean(r64**2))),
                 "rms_err_r_over_rms_r": float(np.sqrt(np.mean((r32 - r64) ** 2)) / np.sqrt(np.mean(r64**2))),
                 "max_abs_err_g_over_rms_g": float(np.max(np.abs(g32 - g64)) / np.sqrt(np.mean(g64**2))),
                 "rms_err_g_over_rms_g": float(np.sqrt(np.mean((g32 - g64) ** 2)) / np.sqrt(np.mean(g64**2)))}
print(json.dumps(out, indent=1))
