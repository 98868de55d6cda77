"""Host handle of the fused HIP iteration object (pg_iter) shared by the FB / FFB iterators."""
import ctypes as C
import warnings
import weakref

from . import _lib
from ._lib import call
from .device import HIPVector


class FusedIteration:
    def __init__(self, f, g, *, fast, Lf, gamma, adaptive, minimum_gamma, reduce_gamma, increase_gamma, mf=0.0,
                 seq_kind=_lib.PG_SEQ_ADAPTIVE, seq_p0=0.0, seq_p1=0.0, reuse_residual=True, single_sweep=True):
        self.f, self.g = f, g
        self.ctx = f.ctx
        o = _lib.pg_iter_opts()
        call("pg_iter_opts_default", C.byref(o))
        o.fast = 1 if fast else 0
        o.adaptive = -1 if adaptive is None else (1 if adaptive else 0)
        o.Lf = float(Lf) if Lf is not None else -1.0
        o.gamma = float(gamma) if gamma is not None else -1.0
        o.minimum_gamma = float(minimum_gamma)
        o.reduce_gamma = float(reduce_gamma)
        o.increase_gamma = float(increase_gamma)
        o.mf = float(mf)
        o.seq_kind = int(seq_kind)
        o.seq_p0, o.seq_p1 = float(seq_p0), float(seq_p1)
        o.g_kind = g.g_kind
        o.g_p0, o.g_p1 = g.g_params()
        o.reuse_residual = 1 if reuse_residual else 0
        o.single_sweep = 1 if single_sweep else 0
        self.opts = o
        h = C.c_void_p()
        call("pg_iter_create", self.ctx.handle, f.handle, C.byref(o), C.byref(h))
        self._h = h
        self._finalizer = weakref.finalize(self, _lib.load().pg_iter_destroy, h)
        self.scalars = _lib.pg_iter_scalars()
        self.n = f.A.n
        self.dtype = f.A.dtype
        self._g_vectors = None

    def _bind_g_vectors(self, like):
        """IndBox with per-element bounds: (lo, hi); NormL1 with per-element weights: (lam, None) --
        pg_iter_set_g_vectors, once; `like` is any device vector of the problem's element type and context"""
        if hasattr(self.g, "g_vectors") and self._g_vectors is None:
            v0, v1 = self.g.g_vectors(like)
            if v0 is not None:
                if v0.n != self.n or (v1 is not None and v1.n != self.n):
                    raise ValueError("per-element parameters of g must have one entry per variable")
                call("pg_iter_set_g_vectors", self._h, v0.vp, v1.vp if v1 is not None else None)
                self._g_vectors = (v0, v1)  # keep the device vectors alive as long as the iterator

    def init(self, x0):
        self._bind_g_vectors(x0)
        call("pg_iter_init", self._h, x0.vp, C.byref(self.scalars))
        return self.scalars

    def step(self, host_beta=0.0):
        call("pg_iter_step", self._h, float(host_beta), C.byref(self.scalars))
        if self.scalars.flags & _lib.PG_FLAG_GAMMA_TOO_SMALL:
            warnings.warn(f"stepsize `gamma` became too small ({self.scalars.gamma})")  # fb_tools.jl:59-61
        return self.scalars

    def run(self, k_start, maxit, tol, check_every=1):
        """IterativeAlgorithm loop inside the library; check_every > 1 (fixed step only) enqueues that many
        iterations between host synchronisations (pg_iter_run_batched)."""
        k = C.c_int64()
        if check_every > 1:
            call("pg_iter_run_batched", self._h, int(k_start), int(maxit), float(tol), int(check_every), C.byref(k),
                 C.byref(self.scalars))
        else:
            call("pg_iter_run", self._h, int(k_start), int(maxit), float(tol), C.byref(k), C.byref(self.scalars))
        return k.value, self.scalars

    def run_small(self, k_start, maxit, tol):
        """Whole solve in one launch of one workgroup (pg_iter_run_small; m * n <= 2^20 elements)."""
        k = C.c_int64()
        call("pg_iter_run_small", self._h, int(k_start), int(maxit), float(tol), C.byref(k), C.byref(self.scalars))
        return k.value, self.scalars

    def run_coop(self, k_start, maxit, tol, blocks=0):
        """Whole solve in one cooperative launch of up to one workgroup per CU (pg_iter_run_coop; cache-resident A,
        m <= 4096 f64 / 8192 f32 rows).  blocks = 0: chosen from the size of A."""
        k = C.c_int64()
        call("pg_iter_run_coop", self._h, int(k_start), int(maxit), float(tol), int(blocks), C.byref(k),
             C.byref(self.scalars))
        return k.value, self.scalars

    def state_download(self):
        """all algorithm memory of the iterator as one host blob (pg_iter_state_download): what `iterate(iter, saved_state)`
        resumes from in the reference (fast_forward_backward.jl:60-71)"""
        nbytes = C.c_int64()
        call("pg_iter_state_bytes", self._h, C.byref(nbytes))
        blob = (C.c_char * nbytes.value)()
        call("pg_iter_state_download", self._h, blob, nbytes.value)
        return bytes(blob)

    def state_upload(self, blob):
        """put a saved state into this (freshly created, same options) iterator; the next step() continues the saved solve"""
        self._bind_g_vectors(HIPVector.empty(self.n, self.dtype, self.ctx))
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        call("pg_iter_state_upload", self._h, buf, len(blob), C.byref(self.scalars))
        return self.scalars

    def view(self):
        st = _lib.pg_iter_state()
        call("pg_iter_state_view", self._h, C.byref(st))
        mk = lambda p: HIPVector(self.ctx, p, self.n, self.dtype, owner=self) if p else None
        return {name: mk(getattr(st, name)) for name, _ in _lib.pg_iter_state._fields_}
