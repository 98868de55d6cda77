"""Re-run single cases of tests/tools/fuzz_newton.py with the objective values printed: python scripts/fuzz_case.py seed..."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import fuzz_newton as fz
import proximalalgorithms.jl_amd as pa
from oracle import proxgrad_oracle as o
pa.get_context()
for seed in [int(s) for s in sys.argv[1:]]:
    print(fz.one_case(seed))
    rng = np.random.default_rng(seed)
    dtype = np.float64 if rng.random() < 0.7 else np.float32
    alg = rng.choice(["panoc", "zerofpr", "panocplus", "dr"])
    m, n = int(rng.integers(2, 300)), int(rng.integers(2, 500))
    A = np.asfortranarray(rng.standard_normal((m, n)).astype(dtype) / dtype(np.sqrt(m)))
    xt = np.zeros(n, dtype); nzc = max(1, n // 10)
    xt[rng.choice(n, nzc, replace=False)] = rng.standard_normal(nzc).astype(dtype)
    b = (A @ xt + dtype(0.01) * rng.standard_normal(m).astype(dtype)).astype(dtype)
    loss = rng.choice(["sqdist", "logistic"])
    L, Lo = (pa.SquaredDistance, o.SquaredDistance) if loss == "sqdist" else (pa.LogisticLoss, o.LogisticLoss)
    lam = dtype(0.1) * dtype(np.max(np.abs(A.T @ b))) if loss == "sqdist" else dtype(0.02)
    box = rng.random() < 0.25
    g_g, g_o = (pa.IndBox(dtype(-0.5), dtype(0.5)), o.IndBox(dtype(-0.5), dtype(0.5))) if box else (pa.NormL1(lam), o.NormL1(lam))
    tol = 1e-4 if dtype == np.float32 else 1e-7
    x0 = np.zeros(n, dtype)
    G, O = {"panoc": (pa.PANOC, o.panoc), "zerofpr": (pa.ZeroFPR, o.zerofpr), "panocplus": (pa.PANOCplus, o.panocplus)}[alg]
    A64, b64 = A.astype(np.float64), b.astype(np.float64)
    def obj(v):
        t = A64 @ v.astype(np.float64) - b64
        fv = 0.5 * np.sum(t * t) if loss == "sqdist" else np.sum(np.log1p(np.exp(-t)))
        return fv + (0.0 if box else float(lam) * np.sum(np.abs(v)))
    for maxit in (50, 100, 200, 236, 400, 800):
        z_o, k_o = O(tol=tol, maxit=maxit, x0=x0, f=Lo(b), A=A, g=g_o)
        z, k = G(tol=tol, maxit=maxit)(x0=x0, f=L(b), A=A, g=g_g)
        print(f"  maxit={maxit}: gpu k={k} F={obj(z):.6e}   cpu k={k_o} F={obj(z_o):.6e}   F(x0)={obj(x0):.4e}  |z-z_o|={np.max(np.abs(z - z_o)):.2e}")
