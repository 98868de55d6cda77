import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proximalalgorithms.jl_amd as pa
from oracle import proxgrad_oracle as o
for (m, n) in ((200, 500), (50, 100)):
    A, b, _ = o.synthetic_lasso(m, n, seed=0, dtype=np.float64 if m < 2000 else np.float32)
    lam = 0.1 * np.max(np.abs(A.T @ b)); Lf = float(np.linalg.norm(A, 2) ** 2)
    f, g = pa.LeastSquares(A, b), pa.NormL1(lam)
    for kw in (dict(Lf=Lf), {}):
        for blocks in (0, 16):
            it = pa.FastForwardBackwardIteration(f=f, g=g, x0=np.zeros(n, A.dtype), **kw); next(iter(it))
            it._fused.run_coop(1, 301, 0.0, blocks)
