"""L-BFGS operator on the device -- mirror of src/accel/lbfgs.jl (LBFGSOperator, update!, reset!, mul!, *)."""
import ctypes as C
import weakref

from . import _lib
from ._lib import call
from .device import pg_dtype


class LBFGSOperator:
    """LBFGSOperator{M}(x)  (lbfgs.jl:5-26): circular (s, y) memory of size M, H = <s,y>/<y,y>."""

    def __init__(self, M, x):
        self.ctx = x.ctx
        self.M, self.n, self.dtype = int(M), x.n, x.dtype
        h = C.c_void_p()
        call("pg_lbfgs_create", self.ctx.handle, pg_dtype(self.dtype), self.M, self.n, C.byref(h))
        self._h = h
        self._finalizer = weakref.finalize(self, _lib.load().pg_lbfgs_destroy, h)
        # update_ calls since the last reset_: 0 means the memory is certainly EMPTY (then mul! is the identity: lbfgs.jl:52-55,
        # 64-71 with currmem = 0 and H = 1); > 0 means "perhaps not" (an update with <s, y> <= 0 is skipped on the device, :33)
        self.updates_since_reset = 0

    def update_(self, s, y):
        """update!(L, s, y)  lbfgs.jl:30-50"""
        call("pg_lbfgs_update", self._h, s.vp, y.vp)
        self.updates_since_reset += 1
        return self

    def reset_(self):
        """reset!(L)  lbfgs.jl:52-55"""
        call("pg_lbfgs_reset", self._h)
        self.updates_since_reset = 0
        return self

    def mul_(self, d, v):
        """mul!(d, L, v)  lbfgs.jl:64-95 (two-loop recursion)"""
        call("pg_lbfgs_apply", self._h, d.vp, v.vp)
        return d

    # ---- images under a linear map (see include/proxgrad_hip.h: pg_lbfgs_images_*) ----
    def images_enable(self, m):
        """keep A s_i, A y_i (m-vectors) next to the stored pairs: pg_lbfgs_images_enable"""
        call("pg_lbfgs_images_enable", self._h, int(m))
        return self

    def images_update_(self, As, Ay):
        """to be called right after update_(s, y) with A s, A y"""
        call("pg_lbfgs_images_update", self._h, As.vp, Ay.vp)
        return self

    def images_mul_(self, Ad, Av):
        """to be called right after mul_(d, v): Ad = A d from A v"""
        call("pg_lbfgs_images_apply", self._h, Ad.vp, Av.vp)
        return Ad

    def images_ready(self):
        """whether every pair the next mul_ will use has its images (then images_mul_ is valid after it)"""
        out = C.c_int32()
        call("pg_lbfgs_images_ready", self._h, C.byref(out))
        return bool(out.value)

    def __mul__(self, v):
        """L * v  lbfgs.jl:57-60"""
        return self.mul_(v.similar(), v)


class LBFGS:
    """LBFGS(M): acceleration-style tag with `initialize` (lbfgs.jl:97-105)."""

    def __init__(self, M):
        self.M = int(M)

    def initialize(self, x):
        return LBFGSOperator(self.M, x)
