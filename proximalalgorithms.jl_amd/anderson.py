"""Anderson acceleration on the device -- mirror of src/accel/anderson.jl.

d = v + (S - Y) pinv(Y'Y) Y'v over a circular memory of M pairs.  The n-vectors and every product with them live on the
device (dots and AXPBYs of the library); the M x M Gram system (M ~ 5) is solved on the host in the working precision,
as the reference does with ``pinv``.  The Gram matrix is kept incrementally: one update costs M dots instead of M^2.
"""
import numpy as np


class AndersonAccelerationOperator:
    """anderson.jl:5-24 (state), update! :26-42, reset! :44-46, mul! :53-62"""

    def __init__(self, M, x):
        self.M = int(M)
        self.currmem = self.curridx = 0  # curridx is 1-based like the reference; 0 = empty
        self.s_M = [x.similar().fill_(0.0) for _ in range(self.M)]
        self.y_M = [x.similar().fill_(0.0) for _ in range(self.M)]
        self.dtype = x.dtype
        self._gram = np.zeros((self.M, self.M), dtype=x.dtype)  # Y'Y over the slots
        self._diff = x.similar()

    def update_(self, s, y):
        """update!(L, s, y)"""
        self.curridx += 1
        if self.curridx > self.M:
            self.curridx = 1
        self.currmem = min(self.currmem + 1, self.M)
        i = self.curridx - 1
        self.s_M[i].copy_from(s)
        self.y_M[i].copy_from(y)
        for j in range(self.currmem):
            self._gram[i, j] = self._gram[j, i] = self.y_M[i].dot(self.y_M[j])
        return self

    def reset_(self):
        """reset!(L)"""
        self.currmem = self.curridx = 0
        return self

    def mul_(self, d, v):
        """mul!(d, L, v)"""
        if d.ptr != v.ptr:
            d.copy_from(v)
        k = self.currmem
        if k == 0:
            return d
        R = self.dtype.type
        Ytv = np.array([self.y_M[j].dot(v) for j in range(k)], dtype=self.dtype)
        c = (np.linalg.pinv(self._gram[:k, :k]) @ Ytv).astype(self.dtype)
        for j in range(k):  # d += c_j (s_j - y_j)
            self._diff.axpby_(1.0, self.s_M[j], -1.0, self.y_M[j])
            d.axpby_(1.0, d, float(R(c[j])), self._diff)
        return d

    def __mul__(self, v):
        """L * v  (anderson.jl:48-51)"""
        return self.mul_(v.similar(), v)


class AndersonAcceleration:
    """AndersonAcceleration(M): quasi-Newton-style tag with `initialize` (anderson.jl:64-72)"""

    def __init__(self, M):
        self.M = int(M)

    def initialize(self, x):
        return AndersonAccelerationOperator(self.M, x)
