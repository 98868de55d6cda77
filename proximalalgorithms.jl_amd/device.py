"""Device context, vectors and the dense matrix object (host-side handles over the C ABI).

PyTorch is used here only as plumbing: device memory (``torch.empty``), the current HIP stream and
``torch.distributed``; every arithmetic operation goes through libproxgrad_hip's kernels.
"""
import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import PG_F32, PG_F64, call

_NP2PG = {np.dtype(np.float32): PG_F32, np.dtype(np.float64): PG_F64}
_PG2NP = {PG_F32: np.dtype(np.float32), PG_F64: np.dtype(np.float64)}


def pg_dtype(dtype):
    dt = np.dtype(dtype)
    if dt not in _NP2PG:
        raise TypeError(f"unsupported element type {dt}; the HIP engine computes in Float32 or Float64")
    return _NP2PG[dt]


class Context:
    """One HIP device + stream + workspace (pg_ctx).  One host thread at a time."""

    def __init__(self, device=None, stream=None):
        import torch

        if not torch.cuda.is_available():
            raise _lib.ProxGradError("no HIP device visible: the proximal-gradient engine needs an MI355X (gfx950)")
        if device is None:
            device = torch.cuda.current_device()
        self.device = int(device)
        self.torch_device = torch.device("cuda", self.device)
        if stream is None:
            stream = torch.cuda.current_stream(self.torch_device).cuda_stream
        self.stream = int(stream)
        h = C.c_void_p()
        call("pg_ctx_create", self.device, C.c_void_p(self.stream), C.byref(h))
        self._h = h
        self._allreduce_cb = None
        self._stream_owner = None
        self.capturing = False
        self._finalizer = weakref.finalize(self, _lib.load().pg_ctx_destroy, h)

    @classmethod
    def on_new_stream(cls, device=None):
        """A context on its own (non-default) HIP stream -- what stream capture needs (the null stream cannot be
        captured).  The torch stream object is kept alive by the context."""
        import torch

        if device is None:
            device = torch.cuda.current_device()
        st = torch.cuda.Stream(device=torch.device("cuda", int(device)))
        ctx = cls(device, st.cuda_stream)
        ctx._stream_owner = st
        return ctx

    @property
    def handle(self):
        return self._h

    def sync(self):
        call("pg_ctx_sync", self._h)

    # ---- stream capture: record a launch-bound iteration body once, replay it with one launch ----
    def capture_begin(self):
        """Start recording: library calls on this context are captured into a hipGraph instead of executed; device
        allocations through HIPVector.empty are refused until capture_end (their addresses would be baked in)."""
        call("pg_ctx_capture_begin", self._h)
        self.capturing = True

    def capture_end(self, abort=False):
        """Stop recording; returns the instantiated :class:`Graph` (None when aborting)."""
        self.capturing = False
        if abort:
            call("pg_ctx_capture_end", self._h, None)
            return None
        g = C.c_void_p()
        call("pg_ctx_capture_end", self._h, C.byref(g))
        return Graph(self, g)

    def device_info(self):
        info = _lib.pg_device_info()
        call("pg_ctx_device_info", self._h, C.byref(info))
        return {"device": info.device, "compute_units": info.compute_units, "wavefront_size": info.wavefront_size,
                "lds_bytes_per_cu": info.lds_bytes_per_cu, "global_mem_bytes": info.global_mem_bytes,
                "clock_khz": info.clock_khz, "arch": info.arch.decode(), "name": info.name.decode()}

    def set_column_sharding(self, nranks, rank):
        """This context's operators hold a COLUMN block of A (see pg_ctx_set_column_sharding); nranks = 0: row sharding."""
        call("pg_ctx_set_column_sharding", self._h, int(nranks), int(rank))

    def profile(self, enable=True, kernels=None):
        """Bracket kernel launches with HIP event pairs; ``kernels``: names from _lib.KERNEL_NAMES to restrict the
        pairs to (every pair is a marker packet on the stream), default all."""
        mask = 0xFFFFFFFF
        if kernels is not None:
            mask = 0
            for name in kernels:
                mask |= 1 << _lib.KERNEL_NAMES.index(name)
        call("pg_ctx_profile_select", self._h, mask)
        call("pg_ctx_profile_enable", self._h, 1 if enable else 0)

    def profile_reset(self):
        call("pg_ctx_profile_reset", self._h)

    def profile_read(self):
        """{kernel name: (launches, total_ms)} measured with HIP events on this context's stream."""
        out = {}
        for k, name in enumerate(_lib.KERNEL_NAMES):
            n, ms = C.c_int64(), C.c_double()
            call("pg_ctx_profile_read", self._h, k, C.byref(n), C.byref(ms))
            out[name] = (n.value, ms.value)
        return out

    def set_allreduce(self, fn):
        """fn(ptr:int, count:int, pg_dtype:int, stream:int) -> None performs an in-place SUM all-reduce on the
        device buffer; None clears it.  Used for row-sharded LeastSquares."""
        if fn is None:
            self._allreduce_cb = None
            call("pg_ctx_set_allreduce", self._h, _lib.ALLREDUCE_FN(), None)
            return

        def _cb(user, buf, count, dtype, stream):
            try:
                fn(int(buf), int(count), int(dtype), int(stream or 0))
                return 0
            except Exception as exc:  # surfaced as PG_ERR_COLLECTIVE
                import traceback

                traceback.print_exc()
                self._allreduce_error = exc
                return 1

        self._allreduce_cb = _lib.ALLREDUCE_FN(_cb)  # keep alive
        call("pg_ctx_set_allreduce", self._h, self._allreduce_cb, None)

    def set_allreduce_async(self, begin, wait):
        """begin(ptr, count, pg_dtype, stream) issues an asynchronous SUM all-reduce ordered after the stream's
        current work; wait(stream) makes the stream wait for all of them (see pg_ctx_set_allreduce_async)."""
        if begin is None:
            self._allreduce_async_cbs = None
            call("pg_ctx_set_allreduce_async", self._h, _lib.ALLREDUCE_FN(), _lib.ALLREDUCE_WAIT_FN(), None)
            return

        def _b(user, buf, count, dtype, stream):
            try:
                begin(int(buf), int(count), int(dtype), int(stream or 0))
                return 0
            except Exception:
                import traceback

                traceback.print_exc()
                return 1

        def _w(user, stream):
            try:
                wait(int(stream or 0))
                return 0
            except Exception:
                import traceback

                traceback.print_exc()
                return 1

        self._allreduce_async_cbs = (_lib.ALLREDUCE_FN(_b), _lib.ALLREDUCE_WAIT_FN(_w))
        call("pg_ctx_set_allreduce_async", self._h, self._allreduce_async_cbs[0], self._allreduce_async_cbs[1], None)


class Graph:
    """An instantiated hipGraph of recorded library calls (pg_graph); ``launch()`` replays it on the context's stream."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self._h = handle
        self._finalizer = weakref.finalize(self, _lib.load().pg_graph_destroy, handle)

    def launch(self):
        call("pg_graph_launch", self._h)


_default_ctx = {}


def set_default_context(ctx):
    """Make ``ctx`` the process-wide default of its device (vectors / operators created without an explicit context
    use it).  Returns the previous default (or None)."""
    prev = _default_ctx.get(ctx.device)
    _default_ctx[ctx.device] = ctx
    return prev


def get_context(device=None):
    """Process-wide default context of a device (created on first use with torch's current stream)."""
    import torch

    if device is None:
        if not torch.cuda.is_available():
            raise _lib.ProxGradError("no HIP device visible: the proximal-gradient engine needs an MI355X (gfx950)")
        device = torch.cuda.current_device()
    device = int(device)
    if device not in _default_ctx:
        # PROXGRAD_SIDE_STREAM=1: the default context runs on its own (capturable) stream instead of torch's current one
        import os

        _default_ctx[device] = Context.on_new_stream(device) if os.environ.get("PROXGRAD_SIDE_STREAM") == "1" else Context(device)
    return _default_ctx[device]


class _RawDeviceArray:
    """__cuda_array_interface__ view of a raw pointer (lets torch wrap library-owned memory)."""

    def __init__(self, ptr, n, np_dtype, owner):
        self.owner = owner
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": np.dtype(np_dtype).str,
                                         "data": (int(ptr), False), "version": 2, "strides": None}


class HIPVector:
    """A device n-vector of Float32/Float64 (the ``Tx`` of the reference's iterators).

    Owns nothing itself: ``owner`` keeps the backing allocation alive (a torch tensor for vectors
    allocated here, the iterator object for views of library-owned state)."""

    __slots__ = ("ctx", "ptr", "n", "dtype", "owner", "__weakref__")

    def __init__(self, ctx, ptr, n, dtype, owner=None):
        self.ctx = ctx
        self.ptr = int(ptr) if ptr else 0
        self.n = int(n)
        self.dtype = np.dtype(dtype)
        self.owner = owner

    # ---- construction (Julia: similar / zero / copy / Array(x)) ----
    @classmethod
    def empty(cls, n, dtype, ctx=None):
        import torch

        ctx = ctx or get_context()
        if ctx.capturing:
            raise _lib.ProxGradError("device allocation during stream capture: the recorded body must reuse buffers "
                                     "allocated by an earlier, uncaptured run")
        tdt = torch.float32 if np.dtype(dtype) == np.float32 else torch.float64
        pg_dtype(dtype)
        t = torch.empty(max(int(n), 1), dtype=tdt, device=ctx.torch_device)
        return cls(ctx, t.data_ptr(), n, dtype, owner=t)

    @classmethod
    def zeros(cls, n, dtype, ctx=None):
        v = cls.empty(n, dtype, ctx)
        call("pg_memset_zero", v.ctx.handle, C.c_void_p(v.ptr), v.nbytes)
        return v

    @classmethod
    def from_numpy(cls, arr, ctx=None):
        arr = np.ascontiguousarray(arr)
        if arr.ndim != 1:
            raise ValueError("HIPVector.from_numpy expects a 1-D array")
        v = cls.empty(arr.shape[0], arr.dtype, ctx)
        if v.n:
            call("pg_memcpy_h2d", v.ctx.handle, C.c_void_p(v.ptr), arr.ctypes.data_as(C.c_void_p), v.nbytes)
        return v

    @classmethod
    def from_torch(cls, t, ctx=None):
        import torch

        if t.dim() != 1 or not t.is_contiguous() or not t.is_cuda:
            raise ValueError("expected a contiguous 1-D device tensor")
        dt = {torch.float32: np.float32, torch.float64: np.float64}[t.dtype]
        ctx = ctx or get_context(t.device.index)
        return cls(ctx, t.data_ptr(), t.shape[0], dt, owner=t)

    @property
    def nbytes(self):
        return self.n * self.dtype.itemsize

    @property
    def pg_dtype(self):
        return pg_dtype(self.dtype)

    @property
    def vp(self):
        return C.c_void_p(self.ptr)

    def __len__(self):
        return self.n

    def similar(self):
        return HIPVector.empty(self.n, self.dtype, self.ctx)

    def copy(self):
        out = self.similar()
        out.copy_from(self)
        return out

    def copy_from(self, other):
        """copyto!(self, other)"""
        if isinstance(other, HIPVector):
            _check_same(self, other)
            call("pg_memcpy_d2d", self.ctx.handle, self.vp, other.vp, self.nbytes)
        else:
            arr = np.ascontiguousarray(other, dtype=self.dtype)
            if arr.shape != (self.n,):
                raise ValueError("shape mismatch")
            if self.n:
                call("pg_memcpy_h2d", self.ctx.handle, self.vp, arr.ctypes.data_as(C.c_void_p), self.nbytes)
        return self

    def numpy(self):
        out = np.empty(self.n, dtype=self.dtype)
        if self.n:
            call("pg_memcpy_d2h", self.ctx.handle, out.ctypes.data_as(C.c_void_p), self.vp, self.nbytes)
        return out

    def torch(self):
        """Zero-copy torch view (used for collectives)."""
        import torch

        if self.owner is not None and isinstance(self.owner, torch.Tensor) and self.owner.data_ptr() == self.ptr:
            return self.owner[: self.n]
        return torch.as_tensor(_RawDeviceArray(self.ptr, self.n, self.dtype, self.owner), device=self.ctx.torch_device)

    # ---- BLAS-1 (all through the C ABI) ----
    def fill_(self, c):
        call("pg_fill", self.ctx.handle, self.pg_dtype, self.n, self.vp, float(c))
        return self

    def axpby_(self, a, x, b=0.0, y=None):
        """self .= a .* x .+ b .* y"""
        _check_same(self, x)
        if y is not None:
            _check_same(self, y)
        call("pg_axpby", self.ctx.handle, self.pg_dtype, self.n, self.vp, float(a), x.vp, float(b),
             y.vp if y is not None else None)
        return self

    def add_scalar_(self, x, c):
        _check_same(self, x)
        call("pg_add_scalar", self.ctx.handle, self.pg_dtype, self.n, self.vp, x.vp, float(c))
        return self

    def dot(self, other):
        _check_same(self, other)
        out = C.c_double()
        call("pg_dot", self.ctx.handle, self.pg_dtype, self.n, self.vp, other.vp, C.byref(out))
        return self.dtype.type(out.value)

    def norm(self):
        out = C.c_double()
        call("pg_nrm2sq", self.ctx.handle, self.pg_dtype, self.n, self.vp, C.byref(out))
        return self.dtype.type(np.sqrt(self.dtype.type(max(out.value, 0.0))))

    def norm_inf(self):
        out = C.c_double()
        call("pg_nrminf", self.ctx.handle, self.pg_dtype, self.n, self.vp, C.byref(out))
        return self.dtype.type(out.value)

    def __repr__(self):
        return f"HIPVector(n={self.n}, dtype={self.dtype}, device={self.ctx.device})"


def _check_same(a, b):
    if a.n != b.n or a.dtype != b.dtype:
        raise ValueError(f"vector mismatch: ({a.n}, {a.dtype}) vs ({b.n}, {b.dtype})")


def as_hipvector(x, ctx=None):
    if isinstance(x, HIPVector):
        return x
    try:
        import torch

        if isinstance(x, torch.Tensor):
            return HIPVector.from_torch(x, ctx)
    except ImportError:  # pragma: no cover
        pass
    return HIPVector.from_numpy(np.asarray(x), ctx)


class HIPMatrix:
    """Dense column-major m x n matrix on the device (Julia ``Matrix{T}``), library-owned, leading
    dimension padded to 1 KiB (pg_mat)."""

    def __init__(self, m, n, dtype, ctx=None):
        self.ctx = ctx or get_context()
        self.m, self.n = int(m), int(n)
        self.dtype = np.dtype(dtype)
        h = C.c_void_p()
        call("pg_mat_create", self.ctx.handle, pg_dtype(dtype), self.m, self.n, C.byref(h))
        self._h = h
        self._finalizer = weakref.finalize(self, _lib.load().pg_mat_destroy, h)

    @property
    def handle(self):
        return self._h

    @property
    def shape(self):
        return (self.m, self.n)

    @classmethod
    def from_numpy(cls, A, ctx=None):
        A = np.asarray(A)
        if A.ndim != 2:
            raise ValueError("expected a 2-D array")
        Af = np.asfortranarray(A)  # Julia layout
        M = cls(Af.shape[0], Af.shape[1], Af.dtype, ctx)
        if Af.size:
            call("pg_mat_upload", M._h, Af.ctypes.data_as(C.c_void_p), max(Af.shape[0], 1))
        return M

    @classmethod
    def synthetic(cls, m, n, dtype=np.float32, seed=0, row_offset=0, m_global=None, ctx=None, col_offset=0):
        """SURVEY 8(d) instance, generated on the device; bit-identical to the oracle's generator.  (row_offset,
        col_offset): position of this m x n block in the global matrix (row / column shards)."""
        import math

        m_global = m if m_global is None else m_global
        ih8_std = 65536.0 * math.sqrt(8.0 / 12.0) * math.sqrt(1.0 - 1.0 / 65536.0**2)
        scale = float(np.float32(1.0 / (ih8_std * math.sqrt(m_global))))
        M = cls(m, n, dtype, ctx)
        call("pg_mat_generate_block", M._h, C.c_uint32(seed & 0xFFFFFFFF), int(row_offset), int(col_offset), scale)
        return M

    def info(self):
        m, n, ld, dt, p = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32(), C.c_void_p()
        call("pg_mat_info", self._h, C.byref(m), C.byref(n), C.byref(ld), C.byref(dt), C.byref(p))
        return {"m": m.value, "n": n.value, "ld": ld.value, "dtype": dt.value, "ptr": p.value}

    def numpy(self, out=None):
        """the matrix on the host (column-major); `out`: a caller-prepared (m, n) Fortran-ordered array of the same dtype"""
        if out is None:
            out = np.empty((self.m, self.n), dtype=self.dtype, order="F")
        elif out.shape != (self.m, self.n) or out.dtype != self.dtype or not out.flags.f_contiguous:
            raise ValueError("out must be a column-major (m, n) array of the matrix's dtype")
        if out.size:
            call("pg_mat_download", self._h, out.ctypes.data_as(C.c_void_p), max(self.m, 1))
        return out

    def mul(self, x, out=None):
        """mul!(out, A, x)"""
        out = out if out is not None else HIPVector.empty(self.m, self.dtype, self.ctx)
        call("pg_mat_mul", self._h, x.vp, out.vp)
        return out

    def mul_multi(self, xs, outs):
        """outs[k] = A xs[k] for up to three vectors on ONE read of A, each bit-identical to mul's (pg_mat_mul_multi); ProxGradError
        with code PG_ERR_UNSUPPORTED where only single products exist (sharded operators, fewer than 13 row groups)"""
        nv = len(xs)
        xp = (C.c_void_p * nv)(*[v.vp for v in xs])
        yp = (C.c_void_p * nv)(*[v.vp for v in outs])
        call("pg_mat_mul_multi", self._h, nv, xp, yp)
        return outs

    def fused_tn(self, r, x, gamma, g, At_r, y, z, res, Az, image_of_res=False):
        """ONE read of A: At_r = A' r ; y = x - gamma At_r ; z = prox_{gamma g}(y) ; res = x - z ; Az = A z
        (pg_mat_fused_tn) -- or, with image_of_res, Az = A res (pg_mat_fused_tn_res).  Returns (g(z), norm(res, Inf),
        dot(At_r, res), norm(res)^2)."""
        p0, p1 = g.g_params()
        sc = (C.c_double * 4)()
        call("pg_mat_fused_tn_res" if image_of_res else "pg_mat_fused_tn", self._h, r.vp, x.vp, float(gamma), g.g_kind, p0, p1, At_r.vp, y.vp, z.vp, res.vp, Az.vp, sc)
        R = self.dtype.type
        return tuple(R(v) for v in sc)

    def fused_tn_pair(self, r1, x1, r2, x2, gamma, g, out1, out2, image_of_res=False):
        """TWO instances of fused_tn on ONE read of A (pg_mat_fused_tn_pair): out1 / out2 = (At_r, y, z, res, Az) of the pair
        (r1, x1) / (r2, x2); image_of_res: Az = A (x - z) for both (pg_mat_fused_tn_pair_res).  Returns the two scalar quadruples
        (g(z), norm(res, Inf), dot(At_r, res), norm(res)^2)."""
        p0, p1 = g.g_params()
        sc = (C.c_double * 8)()
        call("pg_mat_fused_tn_pair_res" if image_of_res else "pg_mat_fused_tn_pair", self._h, r1.vp, x1.vp, r2.vp, x2.vp, float(gamma), g.g_kind, p0, p1, *[v.vp for v in out1],
             *[v.vp for v in out2], sc)
        R = self.dtype.type
        return tuple(R(v) for v in sc[:4]), tuple(R(v) for v in sc[4:])

    def fused_tn_trio(self, rs, xs, gamma, g, outs, image_of_res=False):
        """THREE instances of fused_tn on ONE read of A (pg_mat_fused_tn_trio): outs[k] = (At_r, y, z, res, Az) of the pair
        (rs[k], xs[k]).  Returns the three scalar quadruples (g(z), norm(res, Inf), dot(At_r, res), norm(res)^2)."""
        p0, p1 = g.g_params()
        sc = (C.c_double * 12)()
        arr = lambda vs: (C.c_void_p * 3)(*[v.vp for v in vs])  # noqa: E731
        call("pg_mat_fused_tn_trio", self._h, arr(rs), arr(xs), float(gamma), g.g_kind, p0, p1, *[arr([o[i] for o in outs]) for i in range(5)],
             1 if image_of_res else 0, sc)
        R = self.dtype.type
        return tuple(tuple(R(v) for v in sc[4 * k:4 * k + 4]) for k in range(3))

    def fused_dys(self, r, xg, z, gamma, relax, g_spec, h_spec, grad, z_half, xh, res, z_next, xg_next, A_xg_next):
        """ONE read of A for a Davis-Yin iteration (pg_mat_fused_dys); g_spec / h_spec = (kind, p0, p1).
        Returns (norm(res, Inf), dot(grad, res), norm(res)^2)."""
        sc = (C.c_double * 4)()
        call("pg_mat_fused_dys", self._h, r.vp, xg.vp, z.vp, float(gamma), float(relax), g_spec[0], float(g_spec[1]),
             float(g_spec[2]), h_spec[0], float(h_spec[1]), float(h_spec[2]), grad.vp, z_half.vp, xh.vp, res.vp, z_next.vp,
             xg_next.vp, A_xg_next.vp, sc)
        R = self.dtype.type
        return R(sc[1]), R(sc[2]), R(sc[3])

    def mul_adjoint(self, r, out=None):
        """mul!(out, A', r)"""
        out = out if out is not None else HIPVector.empty(self.n, self.dtype, self.ctx)
        call("pg_mat_mul_adjoint", self._h, r.vp, out.vp)
        return out
