"""Operator surface of the hot path (drop-in boundary #2).

Mirrors, for device vectors, the three generic functions the reference's iterators call:

  * ``value_and_gradient(f, x) -> (f(x), grad)``      src/ProximalAlgorithms.jl:27-40
  * ``prox_(y, g, x, gamma) -> g(y)``  (``prox!``)    ProximalCore.jl (call sites forward_backward.jl:118 ...)
  * ``prox(g, x, gamma) -> (y, g(y))``                ProximalCore.jl (call sites forward_backward.jl:72 ...)
  * ``gradient_(y, f, x) -> f(x)``     (``gradient!``) ProximalCore <= 0.1 callers

and the operator types on the path: ``LeastSquares`` (with the value_and_gradient method of
benchmark/benchmarks.jl:11-17), ``NormL1``, ``IndBox`` (ProximalOperators.jl) and ``Zero``
(ProximalCore.jl).  Custom operators plug in exactly like in the reference
(docs/src/guide/custom_objectives.jl:13-21): any object with ``value_and_gradient(x)`` /
``prox_(y, x, gamma)`` methods works with the generic iteration path.
"""
import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import PG_G_INDBOX, PG_G_NORML1, PG_G_ZERO, call
from .device import HIPMatrix, HIPVector, as_hipvector


def _few_blas_threads():
    """context manager: cap the host BLAS pool for the small setup factorisations (waking a 64-thread pool costs more
    than a 500 x 500 Cholesky; measured 60-140 ms stalls on the GPU boxes)"""
    try:
        from threadpoolctl import threadpool_limits

        return threadpool_limits(limits=8)
    except Exception:  # threadpoolctl missing: run with the pool as it is
        import contextlib

        return contextlib.nullcontext()


def _gram(A64, left):
    """A A' (left) or A' A in float64 on the host (setup only)"""
    with _few_blas_threads():
        return A64 @ A64.T if left else A64.T @ A64


def _spd_inverse(G):
    """inverse of a symmetric positive definite matrix through its Cholesky factor (float64, host; setup only)"""
    import scipy.linalg as sla

    with _few_blas_threads():
        return sla.cho_solve(sla.cho_factor(G, lower=True, check_finite=False), np.eye(G.shape[0]), check_finite=False)


def is_convex(f):
    """ProximalCore.is_convex(typeof(f)) (trait; operators of this package declare it as a class attribute)"""
    return bool(getattr(f, "is_convex", False))


def is_generalized_quadratic(f):
    """ProximalCore.is_generalized_quadratic(typeof(f))"""
    return bool(getattr(f, "is_generalized_quadratic", False))


class LeastSquares:
    """f(x) = lam/2 ||A x - b||^2 on the device (ProximalOperators.LeastSquares(A, b[, lam])).

    ``A``: numpy 2-D array (uploaded, column-major) or :class:`HIPMatrix`; ``b``: numpy / HIPVector.
    Row-sharded use (one process per GPU): pass the local row block and ``comm`` (see sharding.py);
    every evaluation then all-reduces [grad ; f] over the shards.
    """

    is_convex = True
    is_generalized_quadratic = True

    def __init__(self, A, b, lam=1.0, ctx=None, comm=None):
        if not isinstance(A, HIPMatrix):
            A = HIPMatrix.from_numpy(A, ctx)
        self.A = A
        self.ctx = A.ctx
        self.b = as_hipvector(b, self.ctx)
        if self.b.n != A.m or self.b.dtype != A.dtype:
            raise ValueError(f"b must have length {A.m} and dtype {A.dtype}")
        self.lam = float(lam)
        self.comm = comm
        if comm is not None:
            comm.attach(self.ctx)
        h = C.c_void_p()
        call("pg_ls_create", self.ctx.handle, A.handle, self.b.vp, self.lam, C.byref(h))
        self._h = h
        self._finalizer = weakref.finalize(self, _lib.load().pg_ls_destroy, h)

    @property
    def handle(self):
        return self._h

    @property
    def dtype(self):
        return self.A.dtype

    def value_and_gradient(self, x, out=None):
        """benchmark/benchmarks.jl:11-17: res = A*x - b ; (norm(res)^2 / 2, A' * res)"""
        grad = out if out is not None else HIPVector.empty(self.A.n, self.dtype, self.ctx)
        f = C.c_double()
        call("pg_ls_value_and_gradient", self._h, x.vp, grad.vp, C.byref(f))
        return self.dtype.type(f.value), grad

    def gradient_(self, y, x):
        """ProximalCore.gradient!(y, f, x) -> f(x)"""
        f = C.c_double()
        call("pg_ls_gradient", self._h, y.vp, x.vp, C.byref(f))
        return self.dtype.type(f.value)

    def __call__(self, x):
        f = C.c_double()
        call("pg_ls_value", self._h, x.vp, C.byref(f))
        return self.dtype.type(f.value)

    def fused_pass(self, x, z_old, gamma, beta, g, grad, y, z_new, res, v_next):
        """ONE sweep over A for a whole proximal-gradient iteration (pg_ls_fused_pass): with the residual held by this
        operator being that of ``x``:  grad = lam A' r ; y = x - gamma grad ; z_new = prox_{gamma g}(y) ;
        res = x - z_new ; v_next = z_new + beta (z_new - z_old) ; the residual becomes A v_next - b.
        Returns (f(v_next), g(z_new), norm(res, Inf), dot(grad, res), norm(res)^2)."""
        p0, p1 = g.g_params()
        sc = (C.c_double * 5)()
        call("pg_ls_fused_pass", self._h, x.vp, z_old.vp, float(gamma), float(beta), g.g_kind, p0, p1, grad.vp, y.vp,
             z_new.vp, res.vp, v_next.vp, sc)
        R = self.dtype.type
        return tuple(R(v) for v in sc)

    def prox_(self, y, x, gamma, want_value=True):
        """ProximalCore.prox!(y, f, x, gamma) -> f(y): y = argmin lam/2 ||A z - b||^2 + ||z - x||^2 / (2 gamma)
        = (lam A'A + I/gamma) \\ (lam A'b + x/gamma)  (ProximalOperators.LeastSquares, direct solver).

        Like the reference's cached factorisation, the system matrix is prepared once per gamma on the host (float64
        inverse of the min(m, n)-sized system: normal equations when m >= n, Woodbury when m < n -- meant for the
        small / moderate sizes DouglasRachford-type splittings use it on); every application is device work:
        one or three GEMV passes of the library plus AXPBYs.  Not available on sharded operators."""
        if self.comm is not None:
            raise TypeError("LeastSquares.prox_ is not available on a sharded operator")
        R = self.dtype.type
        gamma = float(R(gamma))
        cache = getattr(self, "_prox_cache", None)
        if cache is None or cache["gamma"] != gamma:
            A64 = self.A.numpy().astype(np.float64)
            m, n = A64.shape
            cache = {"gamma": gamma, "q": x.similar(), "c0": self.A.mul_adjoint(self.b)}
            cache["c0"].axpby_(self.lam, cache["c0"])  # lam A'b
            if m >= n:
                M = _spd_inverse(self.lam * _gram(A64, False) + np.eye(n) / gamma)
                cache["M"] = HIPMatrix.from_numpy(np.asfortranarray(M.astype(self.dtype)), self.ctx)
            else:
                S = _spd_inverse(np.eye(m) + gamma * self.lam * _gram(A64, True))
                cache["S"] = HIPMatrix.from_numpy(np.asfortranarray(S.astype(self.dtype)), self.ctx)
                cache["t"] = HIPVector.empty(m, self.dtype, self.ctx)
                cache["t2"] = HIPVector.empty(m, self.dtype, self.ctx)
                cache["w"] = x.similar()
            self._prox_cache = cache
        q = cache["q"].axpby_(1.0 / gamma, x, 1.0, cache["c0"])  # lam A'b + x / gamma
        if "M" in cache:
            cache["M"].mul(q, y)
        else:
            self.A.mul(q, cache["t"])
            cache["S"].mul(cache["t"], cache["t2"])
            self.A.mul_adjoint(cache["t2"], cache["w"])
            y.axpby_(gamma, q, -gamma * gamma * self.lam, cache["w"])
        return self(y) if want_value else None

    def residual(self):
        """A x - b of the last evaluation (view of the library-owned m-vector)."""
        p = C.c_void_p()
        call("pg_ls_residual_ptr", self._h, C.byref(p))
        return HIPVector(self.ctx, p.value, self.A.m, self.dtype, owner=self)


class NormL1:
    """g(x) = lam ||x||_1 (ProximalOperators.NormL1(lam)); prox = soft threshold.  ``lam`` is a nonnegative scalar or a
    vector of per-element weights (ProximalOperators.NormL1(lambda::AbstractArray): g(x) = sum_i lam_i |x_i|)."""

    g_kind = PG_G_NORML1
    is_convex = True

    def __init__(self, lam=1.0):
        self._scalar = np.isscalar(lam)
        if self._scalar:
            if lam < 0:
                raise ValueError("parameter lam must be nonnegative")
            self.lam = float(lam)
        else:
            self.lam = np.asarray(lam)
            if np.any(self.lam < 0):
                raise ValueError("coefficients in lam must be nonnegative")
        self._lamv = None

    def g_params(self):
        """(lam, 0) as the scalars of the C ABI; per-element weights reach the fused iteration through ``g_vectors``"""
        return (self.lam if self._scalar else 0.0), 0.0

    def g_vectors(self, x):
        """(weights, None) as a device vector shaped like ``x`` ((None, None) for a scalar lam)"""
        return self._weights(x), None

    def _weights(self, x):
        if self._scalar:
            return None
        if self._lamv is None or self._lamv.dtype != x.dtype or self._lamv.ctx is not x.ctx:
            lam = np.ascontiguousarray(np.broadcast_to(self.lam.astype(x.dtype), (x.n,)))
            self._lamv = HIPVector.from_numpy(lam, x.ctx)
        return self._lamv

    def prox_(self, y, x, gamma, want_value=True):
        out = C.c_double() if want_value else None  # no reduction read-back: the call stays asynchronous
        ref = C.byref(out) if want_value else None
        w = self._weights(x)
        if w is not None:
            call("pg_prox_norml1w", x.ctx.handle, x.pg_dtype, x.n, y.vp, x.vp, w.vp, float(gamma), ref)
        else:
            call("pg_prox_norml1", x.ctx.handle, x.pg_dtype, x.n, y.vp, x.vp, self.lam, float(gamma), ref)
        return x.dtype.type(out.value) if want_value else None

    def __call__(self, x):
        out = C.c_double()
        w = self._weights(x)
        if w is not None:
            call("pg_norml1w_value", x.ctx.handle, x.pg_dtype, x.n, x.vp, w.vp, C.byref(out))
        else:
            call("pg_norml1_value", x.ctx.handle, x.pg_dtype, x.n, x.vp, self.lam, C.byref(out))
        return x.dtype.type(out.value)


class IndBox:
    """Indicator of {lo <= x <= hi} (ProximalOperators.IndBox(lo, hi)); prox = projection (clamp).
    Bounds are scalars or vectors."""

    g_kind = PG_G_INDBOX
    is_convex = True

    def __init__(self, lo, hi):
        self.lo, self.hi = lo, hi
        self._scalar = np.isscalar(lo) and np.isscalar(hi)
        if self._scalar and lo > hi:
            raise ValueError("bounds must satisfy lo <= hi")
        self._lov = self._hiv = None

    def g_params(self):
        """(lo, hi) as the scalars of the C ABI; with per-element bounds the fused iteration takes them through
        ``g_vectors`` (pg_iter_set_g_vectors) and ignores these"""
        if not self._scalar:
            return 0.0, 0.0
        return float(self.lo), float(self.hi)

    def g_vectors(self, x):
        """per-element bounds as device vectors shaped like ``x`` (None, None for scalar bounds)"""
        return self._vectors(x)

    def _vectors(self, x):
        if self._scalar:
            return None, None
        if self._lov is None:
            lo = np.broadcast_to(np.asarray(self.lo, dtype=x.dtype), (x.n,))
            hi = np.broadcast_to(np.asarray(self.hi, dtype=x.dtype), (x.n,))
            self._lov, self._hiv = HIPVector.from_numpy(lo, x.ctx), HIPVector.from_numpy(hi, x.ctx)
        return self._lov, self._hiv

    def prox_(self, y, x, gamma):
        lov, hiv = self._vectors(x)
        out = C.c_double()
        call("pg_prox_indbox", x.ctx.handle, x.pg_dtype, x.n, y.vp, x.vp,
             float(self.lo) if self._scalar else 0.0, float(self.hi) if self._scalar else 0.0,
             lov.vp if lov is not None else None, hiv.vp if hiv is not None else None, C.byref(out))
        return x.dtype.type(0)

    def __call__(self, x):
        xs = x.numpy()
        ok = np.all(xs >= np.asarray(self.lo)) and np.all(xs <= np.asarray(self.hi))
        return x.dtype.type(0) if ok else x.dtype.type(np.inf)


class SeparableQuadratic:
    """f(x) = sum_i d_i x_i^2 / 2 + q_i x_i with d >= 0 -- the smooth term of a box-constrained QP with diagonal
    Hessian (ProximalOperators: ``Tilt(SqrNormL2(d), q)`` == ``Quadratic(Diagonal(d), q)``).  ``d``/``q`` are scalars
    or vectors.  prox_{gamma f}(x) = (x - gamma q) ./ (1 + gamma d); value_and_gradient = (f(x), d .* x + q)."""

    is_convex = True  # d >= 0
    is_generalized_quadratic = True

    def __init__(self, d, q, ctx=None):
        self._d_scalar, self._q_scalar = np.isscalar(d), np.isscalar(q)
        self.d = float(d) if self._d_scalar else as_hipvector(d, ctx)
        self.q = float(q) if self._q_scalar else as_hipvector(q, ctx)

    def c_params(self):
        """(d_vec, d, q_vec, q) as the C ABI takes them"""
        return (None if self._d_scalar else self.d.vp, self.d if self._d_scalar else 0.0,
                None if self._q_scalar else self.q.vp, self.q if self._q_scalar else 0.0)

    def prox_(self, y, x, gamma, want_value=True):
        dv, d, qv, q = self.c_params()
        if not want_value:
            call("pg_prox_sepquad", x.ctx.handle, x.pg_dtype, x.n, y.vp, x.vp, dv, d, qv, q, float(gamma), None)
            return None
        out = C.c_double()
        call("pg_prox_sepquad", x.ctx.handle, x.pg_dtype, x.n, y.vp, x.vp, dv, d, qv, q, float(gamma), C.byref(out))
        return x.dtype.type(out.value)

    def __call__(self, x):
        y = x.similar()
        return self.prox_(y, x, 0.0)  # prox with gamma = 0 is the identity and returns f(x)


class _Loss:
    """Smooth loss on an m-vector u (= A x): ``value_and_gradient(u) -> (f(u), grad)``."""

    loss_id = None
    is_generalized_quadratic = False

    def __init__(self, b, ctx=None):
        self.b = as_hipvector(b, ctx)

    def value_and_gradient(self, u, out=None):
        grad = out if out is not None else u.similar()
        f = C.c_double()
        call("pg_loss_value_and_gradient", u.ctx.handle, u.pg_dtype, self.loss_id, u.n, u.vp, self.b.vp, grad.vp,
             C.byref(f))
        return u.dtype.type(f.value), grad

    def __call__(self, u):
        if getattr(self, "_scratch", None) is None or self._scratch.n != u.n:
            self._scratch = u.similar()
        return self.value_and_gradient(u, out=self._scratch)[0]


class SquaredDistance(_Loss):
    """f(u) = ||u - b||^2 / 2 (benchmark/benchmarks.jl:19-28; `x -> norm(x - b)^2 / 2` of
    test_lasso_small.jl:32-33).  Generalized quadratic: PANOC uses the interpolation branch (panoc.jl:215-237)."""

    loss_id = 0
    is_convex = True
    is_generalized_quadratic = True

    def __init__(self, b, lam=1.0, ctx=None):
        super().__init__(b, ctx)
        self.lam = float(lam)

    def value_and_gradient(self, u, out=None):
        v, g = super().value_and_gradient(u, out)
        if self.lam != 1.0:
            g.axpby_(self.lam, g)
            v = u.dtype.type(self.lam) * v
        return v, g

    def prox_(self, y, x, gamma, want_value=True):
        """Translate(SqrNormL2(lam), -b) (test_lasso_small.jl:38, test_elasticnet.jl:24):
        prox = (x + lam gamma b) / (1 + lam gamma); returns f(y)"""
        lg = self.lam * float(gamma)
        y.axpby_(1.0 / (1.0 + lg), x, lg / (1.0 + lg), self.b)
        return self(y) if want_value else None


class SqrNormL2:
    """f(x) = lam/2 ||x||^2 (ProximalOperators.SqrNormL2(lam), test_elasticnet.jl:23): prox = x / (1 + lam gamma),
    gradient lam x."""

    is_convex = True
    is_generalized_quadratic = True

    def __init__(self, lam=1.0):
        if lam < 0:
            raise ValueError("parameter lam must be nonnegative")
        self.lam = float(lam)

    def __call__(self, x):
        R = x.dtype.type
        return R(R(self.lam) / R(2) * x.norm() ** 2)

    def prox_(self, y, x, gamma, want_value=True):
        y.axpby_(1.0 / (1.0 + self.lam * float(gamma)), x)
        return self(y) if want_value else None

    def value_and_gradient(self, x, out=None):
        g = out if out is not None else x.similar()
        g.axpby_(self.lam, x)
        return self(x), g


class Quadratic:
    """f(x) = <x, Q x> / 2 + <q, x> with a dense symmetric Q on the device (ProximalOperators.Quadratic; the closure
    `x -> dot(Q * x, x) / 2 + dot(q, x)` of test_nonconvex_qp.jl:15-18): one GEMV pass per evaluation."""

    def __init__(self, Q, q, ctx=None):
        self.Q = Q if isinstance(Q, HIPMatrix) else HIPMatrix.from_numpy(np.asfortranarray(Q), ctx)
        self.ctx = self.Q.ctx
        self.q = as_hipvector(q, self.ctx)
        self._Qx = HIPVector.empty(self.Q.m, self.Q.dtype, self.ctx)

    def value_and_gradient(self, x, out=None):
        R = x.dtype.type
        self.Q.mul(x, self._Qx)
        v = R(x.dot(self._Qx) / R(2) + self.q.dot(x))
        g = out if out is not None else x.similar()
        g.axpby_(1.0, self._Qx, 1.0, self.q)
        return v, g

    def __call__(self, x):
        R = x.dtype.type
        self.Q.mul(x, self._Qx)
        return R(x.dot(self._Qx) / R(2) + self.q.dot(x))


class Linear:
    """f(x) = <c, x> (ProximalOperators.Linear(c); the closure `x -> dot(c, x)` of test_linear_programs.jl:107):
    gradient c, prox = x - gamma c."""

    is_convex = True
    is_generalized_quadratic = True

    def __init__(self, c, ctx=None):
        self.c = as_hipvector(c, ctx)

    def value_and_gradient(self, x, out=None):
        g = out if out is not None else x.similar()
        g.copy_from(self.c)
        return self.c.dot(x), g

    def prox_(self, y, x, gamma, want_value=True):
        y.axpby_(1.0, x, -float(gamma), self.c)
        return self.c.dot(y) if want_value else None

    def __call__(self, x):
        return self.c.dot(x)


class IndNonnegative(IndBox):
    """indicator of {x >= 0} (ProximalOperators.IndNonnegative): prox = max.(0, x) -- the box kernel with hi = +Inf"""

    def __init__(self):
        super().__init__(0.0, float("inf"))


class IndPoint:
    """indicator of {p} (ProximalOperators.IndPoint(p)): prox = p"""

    is_convex = True

    def __init__(self, p, ctx=None):
        self.p = as_hipvector(p, ctx)

    def prox_(self, y, x, gamma):
        y.copy_from(self.p)
        return x.dtype.type(0)


class IndAffine:
    """indicator of {x : A x = b}, A with full row rank (ProximalOperators.IndAffine(A, b)):
    prox = x - A' (A A')^{-1} (A x - b).  The m x m system is inverted once on the host (float64), every application
    is three GEMV passes and two AXPBYs on the device."""

    is_convex = True

    def __init__(self, A, b, ctx=None):
        self.A = A if isinstance(A, HIPMatrix) else HIPMatrix.from_numpy(np.asfortranarray(A), ctx)
        self.ctx = self.A.ctx
        self.b = as_hipvector(b, self.ctx)
        A64 = self.A.numpy().astype(np.float64)
        S = _spd_inverse(_gram(A64, True))
        self._S = HIPMatrix.from_numpy(np.asfortranarray(S.astype(self.A.dtype)), self.ctx)
        self._t = HIPVector.empty(self.A.m, self.A.dtype, self.ctx)
        self._t2 = HIPVector.empty(self.A.m, self.A.dtype, self.ctx)
        self._w = HIPVector.empty(self.A.n, self.A.dtype, self.ctx)

    def prox_(self, y, x, gamma):
        self.A.mul(x, self._t)
        self._t.axpby_(1.0, self._t, -1.0, self.b)
        self._S.mul(self._t, self._t2)
        self.A.mul_adjoint(self._t2, self._w)
        y.axpby_(1.0, x, -1.0, self._w)
        return x.dtype.type(0)


class SlicedSeparableSum:
    """ProximalOperators.SlicedSeparableSum for contiguous ranges: h(y) = sum_k h_k(y[lo_k:hi_k]); the prox acts on
    device views of the slices (0-based half-open ranges; element offsets must keep 16-byte alignment)."""

    def __init__(self, fs, ranges):
        self.fs, self.ranges = tuple(fs), tuple((int(lo), int(hi)) for lo, hi in ranges)

    @staticmethod
    def _view(v, lo, hi):
        if (lo * v.dtype.itemsize) % 16:
            raise ValueError("slice offsets must be multiples of 16 bytes")
        return HIPVector(v.ctx, v.ptr + lo * v.dtype.itemsize, hi - lo, v.dtype, owner=v)

    def prox_(self, y, x, gamma):
        R = x.dtype.type
        total = R(0)
        for f, (lo, hi) in zip(self.fs, self.ranges):
            total = R(total + f.prox_(self._view(y, lo, hi), self._view(x, lo, hi), gamma))
        return total


class IndZero:
    """ProximalCore.IndZero: indicator of {0}.  Its conjugate is Zero (primal_dual.jl:187 evaluates
    value_and_gradient(convex_conjugate(l), y))."""

    def prox_(self, y, x, gamma):
        y.fill_(0.0)
        return x.dtype.type(0)


class Conjugate:
    """ProximalCore.convex_conjugate(f): prox through Moreau's identity,
    prox_{gamma f*}(x) = x - gamma prox_{f / gamma}(x / gamma); returns f*(y) = <y, p> - f(p)."""

    def __init__(self, f):
        self.f = f
        self._p = self._xs = None

    def prox_(self, y, x, gamma, want_value=True):
        R = x.dtype.type
        if self._p is None or self._p.n != x.n:
            self._p, self._xs = x.similar(), x.similar()
        gamma = float(gamma)
        self._xs.axpby_(1.0 / gamma, x)
        fp = prox_(self._p, self.f, self._xs, 1.0 / gamma, want_value=want_value)
        y.axpby_(1.0, x, -gamma, self._p)
        return R(y.dot(self._p) - fp) if want_value else None


def convex_conjugate(f):
    """ProximalCore.convex_conjugate"""
    if isinstance(f, IndZero):
        return Zero()
    if isinstance(f, Conjugate):
        return f.f
    if isinstance(f, SqrNormL2) and f.lam > 0:
        return SqrNormL2(1.0 / f.lam)  # (lam/2 ||.||^2)* = 1/(2 lam) ||.||^2: smooth, so AFBA's `l` term can use it
    return Conjugate(f)


class LogisticLoss(_Loss):
    """f(u) = sum(log.(1 .+ exp.(-(u .- b)))), labels all one (test_sparse_logistic_small.jl:20-26)."""

    loss_id = 1
    is_convex = True


class Composed:
    """x -> f(A x) as one smooth term (the `fA_autodiff` closures of the reference's tests):
    value_and_gradient(x) = (f(Ax), A' grad f(Ax)) -- two GEMV passes over the device matrix."""

    def __init__(self, f, A, ctx=None):
        self.f = f
        self.A = A if isinstance(A, HIPMatrix) else HIPMatrix.from_numpy(A, ctx)
        self.ctx = self.A.ctx
        self._Ax = HIPVector.empty(self.A.m, self.A.dtype, self.ctx)
        self._gu = HIPVector.empty(self.A.m, self.A.dtype, self.ctx)

    def value_and_gradient(self, x, out=None):
        self.A.mul(x, self._Ax)
        v, gu = self.f.value_and_gradient(self._Ax, out=self._gu)
        return v, self.A.mul_adjoint(gu, out if out is not None else None)

    def __call__(self, x):
        return self.f(self.A.mul(x, self._Ax))


class Zero:
    """ProximalCore.Zero: f(x) = 0; value_and_gradient -> (0, zero(x)) (src/ProximalAlgorithms.jl:38-40);
    prox = identity."""

    g_kind = PG_G_ZERO
    is_convex = True
    is_generalized_quadratic = True

    def g_params(self):
        return 0.0, 0.0

    def value_and_gradient(self, x, out=None):
        g = out if out is not None else x.similar()
        g.fill_(0.0)
        return x.dtype.type(0), g

    def prox_(self, y, x, gamma):
        if y.ptr != x.ptr:
            y.copy_from(x)
        return x.dtype.type(0)

    def __call__(self, x):
        return x.dtype.type(0)


# ---- the generic functions ---------------------------------------------------------------------


def value_and_gradient(f, x):
    """ProximalAlgorithms.value_and_gradient(f, x) -> (f(x), grad f(x))"""
    return f.value_and_gradient(x)


def value_and_gradient_(out, f, x):
    """value_and_gradient with the gradient written into ``out`` (the `state.grad .= grad` pattern of the reference's
    iterators without the intermediate array): returns f(x)."""
    try:
        return f.value_and_gradient(x, out=out)[0]
    except TypeError:  # custom operator without the `out` keyword
        v, g = f.value_and_gradient(x)
        out.copy_from(g)
        return v


def gradient_(y, f, x):
    """ProximalCore.gradient!(y, f, x) -> f(x)"""
    if hasattr(f, "gradient_"):
        return f.gradient_(y, x)
    fx, g = f.value_and_gradient(x)
    y.copy_from(g)
    return fx


def prox_(y, g, x, gamma, want_value=True):
    """ProximalCore.prox!(y, g, x, gamma) -> g(y).  ``want_value=False`` (callers that discard the return value, like
    douglas_rachford.jl:58-60 or primal_dual.jl:186,194) skips the value's reduction and its host read-back where the
    operator supports it; the returned value is then None."""
    if not want_value:
        fn = getattr(g.prox_, "__func__", g.prox_)
        if "want_value" in getattr(getattr(fn, "__code__", None), "co_varnames", ()):
            return g.prox_(y, x, gamma, want_value=False)
        g.prox_(y, x, gamma)  # operator without the keyword: compute the value and drop it
        return None
    return g.prox_(y, x, gamma)


def prox(g, x, gamma):
    """ProximalCore.prox(g, x, gamma) -> (y, g(y))"""
    y = x.similar()
    return y, g.prox_(y, x, gamma)


def fused_supported(f, g):
    """True when (f, g) is the pair the fused HIP iteration is specialised for."""
    return isinstance(f, LeastSquares) and isinstance(g, (NormL1, IndBox, Zero))
