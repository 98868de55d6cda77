"""Broyden (modified, Powell-damped) quasi-Newton operator on the device -- mirror of src/accel/broyden.jl.

H is a dense n x n device matrix (identity at start): applying it is one GEMV pass, updating it two GEMV passes
(H y and s'H) plus the rank-one kernel ``pg_mat_rank1_update`` -- three reads and one write of H per update, all in the
library's HIP kernels.  Memory is n^2 elements, as in the reference (meant for small / moderate n).
"""
import numpy as np

from ._lib import call
from .device import HIPMatrix


class BroydenOperator:
    """broyden.jl:5-16 (H, theta_bar), update! :18-28, reset! :30-33, mul! :40-43"""

    def __init__(self, x, theta_bar=0.2):
        n, R = x.n, x.dtype.type
        self.n, self.dtype, self.ctx = n, x.dtype, x.ctx
        self.theta_bar = R(theta_bar)
        self.H = HIPMatrix.from_numpy(np.asfortranarray(np.eye(n, dtype=x.dtype)), x.ctx)
        self._Hy, self._sH, self._u = x.similar(), x.similar(), x.similar()

    def update_(self, s, y):
        """update!(L, s, y): H += (s - H y) / <s, (1/theta - 1) s + H y> * (s' H)"""
        R = self.dtype.type
        self.H.mul(y, self._Hy)
        self.H.mul_adjoint(s, self._sH)
        ss = R(s.norm() ** 2)
        delta = R(self._Hy.dot(s) / ss)
        if abs(delta) >= self.theta_bar:
            theta = R(1)
        else:
            sgn = R(1) if delta == 0 else R(np.sign(delta))
            theta = R((R(1) - sgn * self.theta_bar) / (R(1) - delta))
        denom = R((R(1) / theta - R(1)) * ss + s.dot(self._Hy))
        self._u.axpby_(1.0, s, -1.0, self._Hy)
        call("pg_mat_rank1_update", self.H.handle, float(R(1) / denom), self._u.vp, self._sH.vp)
        return self

    def reset_(self):
        """reset!(L): H = I"""
        self.H = HIPMatrix.from_numpy(np.asfortranarray(np.eye(self.n, dtype=self.dtype)), self.ctx)
        return self

    def mul_(self, d, v):
        """mul!(d, L, v) = H v"""
        if d.ptr == v.ptr:
            raise ValueError("mul_(d, L, v) needs distinct d and v")
        self.H.mul(v, d)
        return d

    def __mul__(self, v):
        return self.mul_(v.similar(), v)


class Broyden:
    """Broyden(theta_bar = 0.2): quasi-Newton-style tag with `initialize` (broyden.jl:45-53)"""

    def __init__(self, theta_bar=0.2):
        self.theta_bar = theta_bar

    def initialize(self, x):
        return BroydenOperator(x, self.theta_bar)
