"""CPU oracle for the ForwardBackward / FastForwardBackward hot path.

*** TEST INFRASTRUCTURE ONLY ***  Nothing under ``proximalalgorithms.jl_amd/`` may
import this module.  Allowed importers: ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` -- always as the checker / the reported CPU
baseline, never as the product path.

This is a numpy restatement (same operation order, same working precision ``R`` =
dtype of ``x0``) of the reference's Julia code.  Every function cites the
reference ``file:line`` (paths relative to /root/reference) it follows.

Pinning status
--------------
The reference is 100 % Julia and there is no ``julia`` binary in the build image, so
the reference itself cannot be executed here.  The oracle is pinned against the
reference's *own* known answers instead (tests/test_oracle_golden.py):
  * test/problems/test_lasso_small.jl:17-44            (x_star, iteration bounds)
  * test/problems/test_lasso_small_strongly_convex.jl  (x_star, iteration bounds)
  * test/accel/test_lbfgs.jl:6-133                     (dirs_ref golden directions)
  * test/accel/test_nesterov.jl:46-81                  (sequence identities, O(1/k^2))
  * test/utilities/test_fb_tools.jl:19-46              (L_est <= L, monotone backtracking)
  * test/problems/test_equivalence.jl:51-84            (FB iterate sequence, gamma=0.95/L)
  * test/problems/test_nonconvex_qp.jl:33-34           (IndBox prox == clamp)
  * benchmark/data/lasso_{tiny,small,medium}.jld2      (xstar, ystar, KKT with lambda=1)
The arithmetic of ``LeastSquares`` / ``NormL1`` / ``IndBox`` lives in
ProximalOperators.jl (compat "0.15", un-vendored, no Manifest => patch version
unpinned); their published formulas are restated below and independently pinned
by the known answers above.  The one thing no reference *test* exercises is the
``value_and_gradient(::LeastSquaresDirect, x)`` binding itself (only
benchmark/benchmarks.jl:11-17 defines/uses it); it is pinned through the
mathematically identical AutoDifferentiable closure of test_lasso_small.jl:34-35.
"""
from __future__ import annotations

import itertools
import math

import numpy as np

# --------------------------------------------------------------------------------------
# helpers (Julia vector idioms)
# --------------------------------------------------------------------------------------


def _R(x):
    """real(eltype(x)) as a numpy scalar type."""
    return np.asarray(x).dtype.type


def _norm(x):
    """LinearAlgebra.norm(x) in the working precision (BLAS nrm2)."""
    x = np.asarray(x)
    return x.dtype.type(np.linalg.norm(x))


def _norm_inf(x):
    x = np.asarray(x)
    if x.size == 0:
        return x.dtype.type(0)
    return x.dtype.type(np.max(np.abs(x)))


def _dot(x, y):
    x = np.asarray(x)
    return x.dtype.type(np.dot(x, y))


# --------------------------------------------------------------------------------------
# a1: LeastSquares value_and_gradient          benchmark/benchmarks.jl:11-17
# --------------------------------------------------------------------------------------


class LeastSquares:
    """f(x) = lam/2 * ||A x - b||^2  (ProximalOperators.LeastSquares(A, b, lam=1)).

    value_and_gradient follows benchmark/benchmarks.jl:11-17:
        res = f.A * x - f.b ;  norm(res)^2 / 2,  f.A' * res
    (the benchmark binding ignores ``lam``; it is 1 there -- kept general here).
    """

    def __init__(self, A, b, lam=1.0):
        self.A = np.asarray(A)
        self.b = np.asarray(b)
        self.lam = self.A.dtype.type(lam)

    def value_and_gradient(self, x):
        R = _R(x)
        res = self.A @ x - self.b  # benchmarks.jl:15
        val = _norm(res) ** 2 / R(2)  # benchmarks.jl:16
        grad = self.A.T @ res  # benchmarks.jl:16
        if self.lam != 1:
            val = self.lam * val
            grad = self.lam * grad
        return R(val), grad

    def __call__(self, x):
        return self.value_and_gradient(x)[0]

    def prox(self, x, gamma):
        """ProximalOperators.LeastSquares prox (direct solver): argmin_z lam/2 ||Az - b||^2 + ||z - x||^2 / (2 gamma)
        = (A'A + I / (lam gamma)) \\ (A'b + x / (lam gamma)).  Like the reference's operator, the Cholesky factor is
        cached per gamma, and a wide A (m < n) goes through the m x m system (matrix inversion lemma).  Pins the
        DouglasRachford / DRLS restatements against test/problems/test_lasso_small.jl:205-231."""
        import scipy.linalg as sla

        R = _R(x)
        c = R(1) / (R(self.lam) * R(gamma))
        m, n = self.A.shape
        cache = getattr(self, "_chol", None)
        if cache is None or cache[0] != c:
            S = (self.A.T @ self.A if m >= n else self.A @ self.A.T) + c * np.eye(min(m, n), dtype=x.dtype)
            cache = (c, sla.cho_factor(S.astype(x.dtype)), (self.A.T @ self.b).astype(x.dtype))
            self._chol = cache
        q = cache[2] + c * x
        if m >= n:
            y = sla.cho_solve(cache[1], q)
        else:  # (A'A + c I)^-1 q = (q - A' (A A' + c I)^-1 A q) / c
            y = (q - self.A.T @ sla.cho_solve(cache[1], self.A @ q)) / c
        y = y.astype(x.dtype)
        return y, self(y)


class SeparableQuadratic:
    """f(x) = sum_i d_i x_i^2 / 2 + q_i x_i  (ProximalOperators Tilt(SqrNormL2(d), q)):
    prox_{gamma f}(x) = (x - gamma q) ./ (1 + gamma d)."""

    def __init__(self, d, q):
        self.d, self.q = d, q

    def prox(self, x, gamma):
        R = _R(x)
        d = np.asarray(self.d, dtype=x.dtype)
        q = np.asarray(self.q, dtype=x.dtype)
        y = ((x - R(gamma) * q) / (R(1) + R(gamma) * d)).astype(x.dtype)
        return y, self(y)

    def value_and_gradient(self, x):
        d = np.asarray(self.d, dtype=x.dtype)
        q = np.asarray(self.q, dtype=x.dtype)
        return self(x), (d * x + q).astype(x.dtype)

    def __call__(self, x):
        R = _R(x)
        d = np.broadcast_to(np.asarray(self.d, dtype=np.float64), x.shape)
        q = np.broadcast_to(np.asarray(self.q, dtype=np.float64), x.shape)
        x64 = x.astype(np.float64)
        return R(np.sum(0.5 * d * x64 * x64 + q * x64))


class Quadratic:
    """f(x) = <x, Qx>/2 + <q, x>   (ProximalOperators.Quadratic; used by
    test/utilities/test_fb_tools.jl:13)."""

    def __init__(self, Q, q):
        self.Q = np.asarray(Q)
        self.q = np.asarray(q)

    def value_and_gradient(self, x):
        R = _R(x)
        Qx = self.Q @ x
        return R(_dot(x, Qx) / R(2) + _dot(self.q, x)), Qx + self.q

    def __call__(self, x):
        return self.value_and_gradient(x)[0]


class Zero:
    """ProximalCore.Zero: src/ProximalAlgorithms.jl:38-40."""

    def value_and_gradient(self, x):
        return _R(x)(0), np.zeros_like(x)

    def prox(self, x, gamma):
        return x.copy(), _R(x)(0)

    def __call__(self, x):
        return _R(x)(0)


# --------------------------------------------------------------------------------------
# a2: NormL1 prox (soft threshold)             ProximalOperators.jl NormL1 (un-vendored);
#     call sites forward_backward.jl:72,118 ; fast_forward_backward.jl:80,141 ; fb_tools.jl:49
# --------------------------------------------------------------------------------------


class NormL1:
    """g(x) = lam * ||x||_1 ;  prox: y_i = sign(x_i) max(|x_i| - gamma lam, 0); returns g(y).
    ``lam`` may be an array of per-element weights (ProximalOperators.NormL1(lambda::AbstractArray)):
    g(x) = sum_i lam_i |x_i|, threshold gamma lam_i."""

    def __init__(self, lam=1.0):
        self.lam = lam

    def _weighted(self):
        return not np.isscalar(self.lam)

    def prox(self, x, gamma):
        R = _R(x)
        if self._weighted():
            lam = np.asarray(self.lam, dtype=x.dtype)
            gl = (R(gamma) * lam).astype(x.dtype)
            y = np.where(x <= -gl, x + gl, np.where(x >= gl, x - gl, R(0))).astype(x.dtype)
            return y, R(np.sum(lam * np.abs(y), dtype=x.dtype))
        gl = R(gamma) * R(self.lam)
        y = np.where(x <= -gl, x + gl, np.where(x >= gl, x - gl, R(0))).astype(x.dtype)
        return y, R(R(self.lam) * R(np.sum(np.abs(y), dtype=x.dtype)))

    def __call__(self, x):
        R = _R(x)
        if self._weighted():
            return R(np.sum(np.asarray(self.lam, dtype=x.dtype) * np.abs(x), dtype=x.dtype))
        return R(R(self.lam) * R(np.sum(np.abs(x), dtype=x.dtype)))


# --------------------------------------------------------------------------------------
# a3: IndBox prox (projection)                 ProximalOperators.jl IndBox (un-vendored);
#     restated inline by the reference at test/problems/test_nonconvex_qp.jl:33
# --------------------------------------------------------------------------------------


class IndBox:
    """g = indicator of {lo <= x <= hi}; prox = min.(hi, max.(lo, x)); returns 0."""

    def __init__(self, lo, hi):
        self.lo = lo
        self.hi = hi

    def prox(self, x, gamma):
        R = _R(x)
        lo = np.asarray(self.lo, dtype=x.dtype)
        hi = np.asarray(self.hi, dtype=x.dtype)
        return np.minimum(hi, np.maximum(lo, x)).astype(x.dtype), R(0)

    def __call__(self, x):
        R = _R(x)
        ok = np.all(x >= np.asarray(self.lo, dtype=x.dtype)) and np.all(x <= np.asarray(self.hi, dtype=x.dtype))
        return R(0) if ok else R(np.inf)


def value_and_gradient(f, x):
    """src/ProximalAlgorithms.jl:27-40 (generic function; methods on operator types)."""
    return f.value_and_gradient(x)


def prox(g, x, gamma):
    """ProximalCore.prox(g, x, gamma) -> (y, g(y))."""
    return g.prox(x, gamma)


# --------------------------------------------------------------------------------------
# a9/a10: fb_tools                              src/utilities/fb_tools.jl:3-63
# --------------------------------------------------------------------------------------


def f_model(f_x, grad_f_x, res, L):
    """fb_tools.jl:3-5:  f_x - real(dot(grad_f_x, res)) + (L / 2) * norm(res)^2"""
    R = _R(res)
    return R(R(f_x) - _dot(grad_f_x, res) + (R(L) / R(2)) * _norm(res) ** 2)


def lower_bound_smoothness_constant(f, x, grad_f_x=None):
    """fb_tools.jl:7-19 with A = I (as called from forward_backward.jl:70,
    fast_forward_backward.jl:78):
        xeps = x .+ 1 ; grad at xeps ; norm(grad_eps - grad) / sqrt(length(x))
    """
    R = _R(x)
    if grad_f_x is None:  # fb_tools.jl:14-19
        _, grad_f_x = value_and_gradient(f, x)
    xeps = x + R(1)  # :9
    _, grad_eps = value_and_gradient(f, xeps)  # :10
    return R(_norm(grad_eps - grad_f_x) / R(math.sqrt(x.size)))  # :11


def backtrack_stepsize(
    gamma, f, g, x, f_x, grad_f_x, y, z, g_z, res, grad_f_z=None, *, alpha=1.0, minimum_gamma=1e-7, reduce_gamma=0.5,
    counters=None,
):
    """fb_tools.jl:24-63 with A === nothing (Az aliases z).  Arrays y, z, res (and
    grad_f_z when given) are updated in place like the Julia version.
    Returns (gamma, g_z, f_z, f_z_upp)."""
    R = _R(x)
    gamma = R(gamma)
    alpha, minimum_gamma, reduce_gamma = R(alpha), R(minimum_gamma), R(reduce_gamma)
    eps = R(np.finfo(R).eps)
    f_z_upp = f_model(f_x, grad_f_x, res, alpha / gamma)  # :42
    f_z, grad_tmp = value_and_gradient(f, z)  # :44
    tol = R(10) * eps * (R(1) + abs(f_z))  # :45
    nbt = 0
    while f_z > f_z_upp + tol and gamma >= minimum_gamma:  # :46
        gamma = R(gamma * reduce_gamma)  # :47
        y[...] = x - gamma * grad_f_x  # :48
        z_new, g_z = prox(g, y, gamma)  # :49
        z[...] = z_new
        res[...] = x - z  # :50
        f_z_upp = f_model(f_x, grad_f_x, res, alpha / gamma)  # :51
        f_z, grad_tmp = value_and_gradient(f, z)  # :53
        tol = R(10) * eps * (R(1) + abs(f_z))  # :54
        nbt += 1
    if grad_f_z is not None:  # :56-58
        grad_f_z[...] = grad_tmp
    if counters is not None:
        counters["backtracks"] = counters.get("backtracks", 0) + nbt
        counters["gamma_too_small"] = bool(gamma < minimum_gamma)  # :59-61 (warn only)
    return gamma, g_z, f_z, f_z_upp


# --------------------------------------------------------------------------------------
# a11: Nesterov sequences                       src/accel/nesterov.jl
# --------------------------------------------------------------------------------------


def fixed_nesterov_sequence(R):
    """nesterov.jl:14-17: t0 = 1; t+ = (1 + sqrt(1 + 4 t^2)) / 2; yields (t - 1) / t+."""
    R = np.dtype(R).type
    t = R(1)
    while True:
        t_next = R((R(1) + np.sqrt(R(1) + R(4) * t * t)) / R(2))
        yield R((t - R(1)) / t_next)
        t = t_next


def simple_nesterov_sequence(R):
    """nesterov.jl:36: R(k - 1) / (k + 2) for k >= 1."""
    R = np.dtype(R).type
    k = 1
    while True:
        yield R(R(k - 1) / R(k + 2))
        k += 1


def constant_nesterov_sequence(m, stepsize):
    """nesterov.jl:51-54: repeated((1 - sqrt(m s)) / (1 + sqrt(m s)))."""
    R = _R(m)
    k_inverse = R(m) * R(stepsize)
    return itertools.repeat(R((R(1) - np.sqrt(k_inverse)) / (R(1) + np.sqrt(k_inverse))))


class AdaptiveNesterovSequence:
    """nesterov.jl:56-80 (struct) and :89-103 (next!)."""

    def __init__(self, m):
        R = _R(m)
        self.R = R
        self.m = R(m)
        self.stepsize = -R(1)
        self.theta = -R(1)

    def next(self, stepsize):
        R = self.R
        stepsize = R(stepsize)
        if self.stepsize < 0:  # :90-93
            self.stepsize = stepsize
            self.theta = R(np.sqrt(self.m * stepsize)) if self.m > 0 else R(1)
        b = R(self.theta**2 / self.stepsize - self.m)  # :94
        delta = R(b**2 + R(4) * (self.theta**2) / (self.stepsize * stepsize))  # :95
        theta = R(stepsize * (-b + np.sqrt(delta)) / R(2))  # :96
        beta = R(
            stepsize * self.theta * (R(1) - self.theta) / (self.stepsize * theta + stepsize * self.theta**2)
        )  # :97-99
        self.stepsize = stepsize  # :100
        self.theta = theta  # :101
        return beta


# --------------------------------------------------------------------------------------
# a4-a6: ForwardBackward                        src/algorithms/forward_backward.jl:38-129
# --------------------------------------------------------------------------------------


class _State:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class ForwardBackwardIteration:
    """forward_backward.jl:38-48 (options) ; :65-84 (init) ; :86-123 (step)."""

    def __init__(self, *, f=None, g=None, x0, Lf=None, gamma=None, adaptive=None, minimum_gamma=1e-7,
                 reduce_gamma=0.5, increase_gamma=1.0):
        R = _R(x0)
        self.f = f if f is not None else Zero()
        self.g = g if g is not None else Zero()
        self.x0 = x0
        self.Lf = Lf
        self.gamma = gamma if gamma is not None else (None if Lf is None else R(1) / R(Lf))  # :43
        self.adaptive = (self.gamma is None) if adaptive is None else adaptive  # :44
        self.minimum_gamma = R(minimum_gamma)
        self.reduce_gamma = R(reduce_gamma)
        self.increase_gamma = R(increase_gamma)
        self.counters = {}

    def init(self):
        R = _R(self.x0)
        x = self.x0.copy()  # :66
        f_x, grad_f_x = value_and_gradient(self.f, x)  # :67
        if self.gamma is None:  # :68-70
            gamma = R(R(1) / lower_bound_smoothness_constant(self.f, x, grad_f_x))
        else:
            gamma = R(self.gamma)
        y = x - gamma * grad_f_x  # :71
        z, g_z = prox(self.g, y, gamma)  # :72
        return _State(x=x, f_x=R(f_x), grad_f_x=np.array(grad_f_x, copy=True), gamma=gamma, y=y, z=z, g_z=g_z,
                      res=x - z, grad_f_z=np.empty_like(x))  # :73-82

    def step(self, s):
        R = _R(s.x)
        if self.adaptive:  # :90-110
            s.gamma = R(s.gamma * self.increase_gamma)  # :91
            s.gamma, s.g_z, s.f_x, _ = backtrack_stepsize(
                s.gamma, self.f, self.g, s.x, s.f_x, s.grad_f_x, s.y, s.z, s.g_z, s.res, s.grad_f_z,
                minimum_gamma=self.minimum_gamma, reduce_gamma=self.reduce_gamma, counters=self.counters)  # :92-108
            s.x, s.z = s.z, s.x  # :109
            s.grad_f_x, s.grad_f_z = s.grad_f_z, s.grad_f_x  # :110
        else:  # :111-115
            s.x, s.z = s.z, s.x
            s.f_x, grad = value_and_gradient(self.f, s.x)
            s.grad_f_x[...] = grad
        s.y[...] = s.x - s.gamma * s.grad_f_x  # :117
        z_new, s.g_z = prox(self.g, s.y, s.gamma)  # :118
        s.z[...] = z_new
        s.res[...] = s.x - s.z  # :120
        return s

    def __iter__(self):
        s = self.init()
        yield s
        while True:
            yield self.step(s)


# --------------------------------------------------------------------------------------
# a7-a8: FastForwardBackward                    src/algorithms/fast_forward_backward.jl:44-154
# --------------------------------------------------------------------------------------


class FastForwardBackwardIteration:
    """fast_forward_backward.jl:44-56 (options) ; :73-97 (init) ; :106-145 (step)."""

    def __init__(self, *, f=None, g=None, x0, mf=0.0, Lf=None, gamma=None, adaptive=None, minimum_gamma=1e-7,
                 reduce_gamma=0.5, increase_gamma=1.0, extrapolation_sequence=None):
        R = _R(x0)
        self.f = f if f is not None else Zero()
        self.g = g if g is not None else Zero()
        self.x0 = x0
        self.mf = R(mf)
        self.Lf = Lf
        self.gamma = gamma if gamma is not None else (None if Lf is None else R(1) / R(Lf))  # :50
        self.adaptive = (self.gamma is None) if adaptive is None else adaptive  # :51
        self.minimum_gamma = R(minimum_gamma)
        self.reduce_gamma = R(reduce_gamma)
        self.increase_gamma = R(increase_gamma)
        self.extrapolation_sequence = extrapolation_sequence
        self.counters = {}

    def init(self):
        R = _R(self.x0)
        x = self.x0.copy()  # :74
        f_x, grad_f_x = value_and_gradient(self.f, x)  # :75
        if self.gamma is None:  # :76-78
            gamma = R(R(1) / lower_bound_smoothness_constant(self.f, x, grad_f_x))
        else:
            gamma = R(self.gamma)
        y = x - gamma * grad_f_x  # :79
        z, g_z = prox(self.g, y, gamma)  # :80
        if self.extrapolation_sequence is not None:  # :90-94
            seq = iter(self.extrapolation_sequence)
        else:
            seq = AdaptiveNesterovSequence(self.mf)
        return _State(x=x, f_x=R(f_x), grad_f_x=np.array(grad_f_x, copy=True), gamma=gamma, y=y, z=z, g_z=g_z,
                      res=x - z, z_prev=x.copy(), extrapolation_sequence=seq, beta=R(0))  # :81-95 ; z_prev :69

    def step(self, s):
        R = _R(s.x)
        if self.adaptive:  # :110-129  (gradient at z is discarded: grad_f_Az = nothing)
            s.gamma = R(s.gamma * self.increase_gamma)  # :111
            s.gamma, s.g_z, _, _ = backtrack_stepsize(
                s.gamma, self.f, self.g, s.x, s.f_x, s.grad_f_x, s.y, s.z, s.g_z, s.res, None,
                minimum_gamma=self.minimum_gamma, reduce_gamma=self.reduce_gamma, counters=self.counters)
        else:
            s.gamma = R(self.gamma)  # :131
        if isinstance(s.extrapolation_sequence, AdaptiveNesterovSequence):  # :99-104
            beta = s.extrapolation_sequence.next(s.gamma)
        else:
            beta = R(next(s.extrapolation_sequence))
        s.beta = beta
        s.x[...] = s.z + beta * (s.z - s.z_prev)  # :135
        s.z_prev, s.z = s.z, s.z_prev  # :136
        s.f_x, grad = value_and_gradient(self.f, s.x)  # :138
        s.grad_f_x[...] = grad  # :139
        s.y[...] = s.x - s.gamma * s.grad_f_x  # :140
        z_new, s.g_z = prox(self.g, s.y, s.gamma)  # :141
        s.z[...] = z_new
        s.res[...] = s.x - s.z  # :142
        return s

    def __iter__(self):
        s = self.init()
        yield s
        while True:
            yield self.step(s)


# --------------------------------------------------------------------------------------
# a12/a13: stopping rule + driver loop          forward_backward.jl:125-129 ;
#                                               src/ProximalAlgorithms.jl:114-123
# --------------------------------------------------------------------------------------


def default_stopping_criterion(tol, state):
    """norm(state.res, Inf) / state.gamma <= tol   (forward_backward.jl:125-126)."""
    return _norm_inf(state.res) / state.gamma <= tol


def run(iteration, *, maxit=10_000, tol=1e-8, stop=None, trace=None):
    """IterativeAlgorithm call, ProximalAlgorithms.jl:114-123: enumerate from k = 1 (the
    init state); return (solution = state.z, k) when k >= maxit or stop(state)."""
    R = _R(iteration.x0)
    tol = R(tol)
    for k, state in enumerate(iteration, start=1):
        if trace is not None:
            trace.append(dict(k=k, gamma=float(state.gamma), f_x=float(state.f_x), g_z=float(state.g_z),
                              res_inf=float(_norm_inf(state.res)), beta=float(getattr(state, "beta", 0.0)),
                              z=state.z.copy()))
        done = stop(iteration, state) if stop is not None else default_stopping_criterion(tol, state)
        if k >= maxit or done:
            return state.z, k


def douglas_rachford(*, maxit=1_000, tol=1e-8, **kw):
    """DouglasRachford(; maxit, tol)(; kwargs...)  douglas_rachford.jl:101-119; stop :65-69; solution :70 (state.y)."""
    it = DouglasRachfordIteration(**kw)
    R = _R(it.x0)
    for k, s in enumerate(it, start=1):
        if k >= maxit or _norm_inf(s.res) / it.gamma <= R(tol):
            return s.y, k


def forward_backward(*, maxit=10_000, tol=1e-8, trace=None, **kw):
    """ForwardBackward(; maxit, tol)(; kwargs...)   forward_backward.jl:161-179."""
    return run(ForwardBackwardIteration(**kw), maxit=maxit, tol=tol, trace=trace)


def fast_forward_backward(*, maxit=10_000, tol=1e-8, trace=None, **kw):
    """FastForwardBackward(; maxit, tol)(; kwargs...)   fast_forward_backward.jl:186-204."""
    return run(FastForwardBackwardIteration(**kw), maxit=maxit, tol=tol, trace=trace)


# --------------------------------------------------------------------------------------
# a14: L-BFGS operator                          src/accel/lbfgs.jl:5-95
# --------------------------------------------------------------------------------------


class LBFGSOperator:
    def __init__(self, M, x):
        R = _R(x)
        self.M = M
        self.currmem = 0
        self.curridx = 0  # 1-based like the reference; 0 = empty
        self.s = np.zeros_like(x)
        self.y = np.zeros_like(x)
        self.s_M = [np.zeros_like(x) for _ in range(M)]
        self.y_M = [np.zeros_like(x) for _ in range(M)]
        self.ys_M = np.zeros(M, dtype=R)
        self.alphas = np.zeros(M, dtype=R)
        self.H = R(1)

    def update(self, s, y):
        """lbfgs.jl:30-50"""
        self.s[...] = s
        self.y[...] = y
        ys = _dot(self.s, self.y)
        if ys > 0:
            self.curridx += 1
            if self.curridx > self.M:
                self.curridx = 1
            self.currmem = min(self.currmem + 1, self.M)
            self.ys_M[self.curridx - 1] = ys
            self.s_M[self.curridx - 1][...] = self.s
            self.y_M[self.curridx - 1][...] = self.y
            yty = _dot(self.y, self.y)
            self.H = _R(self.s)(ys / yty)
        return self

    def reset(self):
        """lbfgs.jl:52-55"""
        self.currmem = 0
        self.curridx = 0
        self.H = _R(self.s)(1)

    def mul(self, d, v):
        """lbfgs.jl:64-95 two-loop recursion, d <- H_k v."""
        R = _R(d)
        d[...] = v
        idx = self.curridx
        for _ in range(self.currmem):  # loop1! :72-83
            self.alphas[idx - 1] = _dot(self.s_M[idx - 1], d) / self.ys_M[idx - 1]
            d -= self.alphas[idx - 1] * self.y_M[idx - 1]
            idx -= 1
            if idx == 0:
                idx = self.M
        d *= R(self.H)  # :67
        for _ in range(self.currmem):  # loop2! :85-95
            idx += 1
            if idx > self.M:
                idx = 1
            beta = _dot(self.y_M[idx - 1], d) / self.ys_M[idx - 1]
            d += (self.alphas[idx - 1] - beta) * self.s_M[idx - 1]
        return d

    def __mul__(self, v):
        """lbfgs.jl:57-60"""
        return self.mul(np.empty_like(v), v)


# --------------------------------------------------------------------------------------
# "next" rows: DouglasRachford                  src/algorithms/douglas_rachford.jl:30-70
# --------------------------------------------------------------------------------------


class DouglasRachfordIteration:
    """douglas_rachford.jl:30-41 (options), :53-63 (iterate):
        y, = prox!(f, x, gamma); r = 2y - x; z = prox!(g, r, gamma); res = y - z; x -= res
    Stop (:65-69): norm(res, Inf) / gamma <= tol.  Solution (:70): state.y."""

    def __init__(self, *, f, g, x0, gamma):
        self.f, self.g, self.x0 = f, g, x0
        self.gamma = _R(x0)(gamma)

    def __iter__(self):
        s = _State(x=self.x0.copy(), y=np.empty_like(self.x0), r=np.empty_like(self.x0), z=np.empty_like(self.x0),
                   res=np.empty_like(self.x0), gamma=self.gamma)
        while True:
            y, _ = prox(self.f, s.x, self.gamma)
            s.y[...] = y
            s.r[...] = 2 * s.y - s.x
            z, _ = prox(self.g, s.r, self.gamma)
            s.z[...] = z
            s.res[...] = s.y - s.z
            s.x -= s.res
            yield s


# --------------------------------------------------------------------------------------
# synthetic LASSO instance; the reference ships no generator.  The same
# counter-based generator is implemented on the device (csrc/pg_kernels.hip,
# generate_kernel); it uses integer hashing plus ONE int->float conversion and ONE
# float multiply, so host and every GPU row-shard produce bit-identical entries.
# --------------------------------------------------------------------------------------

_M32 = np.uint64(0xFFFFFFFF)
IH8_STD = 65536.0 * math.sqrt(8.0 / 12.0) * math.sqrt(1.0 - 1.0 / 65536.0**2)  # std of the 8-term sum


def _mix32(h):
    """murmur3 fmix32 on uint64 arrays holding 32-bit values."""
    h = h & _M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & _M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & _M32
    h ^= h >> np.uint64(16)
    return h


def counter_ih8(seed, i, j):
    """Integer s(seed, i, j) in [-262140, 262140]: centred sum of eight hashed 16-bit
    uniforms (Irwin-Hall(8): approximately normal, exactly reproducible).
    i, j: broadcastable integer arrays (< 2^32): global row / column index."""
    i = np.asarray(i, dtype=np.uint64) & _M32
    j = np.asarray(j, dtype=np.uint64) & _M32
    h = _mix32(np.uint64((seed & 0xFFFFFFFF) ^ 0x9E3779B9))
    h = _mix32(h ^ j)
    h = _mix32(h ^ ((i * np.uint64(0x9E3779B1)) & _M32))
    s = np.zeros(np.broadcast(i, j).shape, dtype=np.int64)
    for t in range(4):
        w = _mix32((h + np.uint64((t * 0x632BE5AB) & 0xFFFFFFFF)) & _M32)
        s += (w & np.uint64(0xFFFF)).astype(np.int64) + (w >> np.uint64(16)).astype(np.int64)
    return s - 4 * 65535


def synthetic_scale(m_global):
    """float32 multiplier turning counter_ih8 integers into ~N(0,1)/sqrt(m_global)."""
    return np.float32(1.0 / (IH8_STD * math.sqrt(m_global)))


def synthetic_matrix(m, n, seed=0, dtype=np.float32, row_offset=0, m_global=None):
    """A[i, j] = float32(s(seed, row_offset+i, j)) * scale, column-major (Julia layout)."""
    m_global = m if m_global is None else m_global
    scale = synthetic_scale(m_global)
    A = np.empty((m, n), dtype=dtype, order="F")
    rows = np.arange(row_offset, row_offset + m, dtype=np.uint64)[:, None]
    step = max(1, (1 << 21) // max(m, 1))
    for j0 in range(0, n, step):
        j1 = min(n, j0 + step)
        cols = np.arange(j0, j1, dtype=np.uint64)[None, :]
        A[:, j0:j1] = (counter_ih8(seed, rows, cols).astype(np.float32) * scale).astype(dtype)
    return A


def synthetic_lasso(m, n, seed=0, dtype=np.float32, row_offset=0, m_global=None, A=None):
    """SURVEY 8(d) instance: A as above; x_true with max(1, n // 1000) non-zeros ~ N(0,1) at
    seeded positions; b = A x_true + 0.01 N(0,1); lam = 0.1 ||A'b||_inf
    (test/problems/test_lasso_small.jl:29); x0 = 0 (benchmark/benchmarks.jl:58).
    For a row shard pass row_offset/m_global; b and lam are then for the shard's rows only
    (callers combine)."""
    m_global = m if m_global is None else m_global
    if A is None:
        A = synthetic_matrix(m, n, seed, dtype, row_offset, m_global)
    rng = np.random.default_rng(seed + 12345)
    k = max(1, n // 1000)
    x_true = np.zeros(n, dtype=dtype)
    pos = rng.choice(n, size=k, replace=False)
    x_true[pos] = rng.standard_normal(k).astype(dtype)
    noise_all = np.random.default_rng(seed + 54321).standard_normal(m_global).astype(dtype)
    noise = noise_all[row_offset:row_offset + m]
    b = (A @ x_true + dtype(0.01) * noise).astype(dtype)
    return A, b, x_true


# --------------------------------------------------------------------------------------
# "next" rows: PANOC                            src/algorithms/panoc.jl:39-255
# (BASELINE config 4; smooth terms as the reference's tests define them)
# --------------------------------------------------------------------------------------


class SquaredDistance:
    """f(u) = ||u - b||^2 / 2  -- benchmark/benchmarks.jl:19-28 (SquaredDistance) and the closure
    `x -> norm(x - b)^2 / 2` of test/problems/test_lasso_small.jl:32-33.  Generalized quadratic."""

    is_generalized_quadratic = True

    def __init__(self, b):
        self.b = np.asarray(b)

    def value_and_gradient(self, u):
        R = _R(u)
        diff = u - self.b
        return R(_norm(diff) ** 2 / R(2)), diff

    def __call__(self, u):
        return self.value_and_gradient(u)[0]


class LogisticLoss:
    """f(u) = sum(log.(1 .+ exp.(-(u .- b))))  -- test/problems/test_sparse_logistic_small.jl:20-26
    (labels are assumed all one); gradient -1 ./ (1 .+ exp.(u .- b))."""

    is_generalized_quadratic = False

    def __init__(self, b):
        self.b = np.asarray(b)

    def value_and_gradient(self, u):
        R = _R(u)
        t = u - self.b
        val = R(np.sum(np.log(R(1) + np.exp(-t)), dtype=u.dtype))
        grad = (-R(1) / (R(1) + np.exp(t))).astype(u.dtype)
        return val, grad

    def __call__(self, u):
        return self.value_and_gradient(u)[0]


class Composed:
    """x -> f(A x): value_and_gradient = (f(Ax), A' grad f(Ax)); the `fA_autodiff` closures of the tests."""

    def __init__(self, f, A):
        self.f, self.A = f, np.asarray(A)

    def value_and_gradient(self, x):
        v, gu = self.f.value_and_gradient(self.A @ x)
        return v, self.A.T @ gu

    def __call__(self, x):
        return self.f(self.A @ x)


def lower_bound_smoothness_constant_A(f, A, x, grad_f_Ax):
    """fb_tools.jl:7-12 with a general linear map A:
    xeps = x .+ 1 ; grad f at A*xeps ; norm(A' * (grad_eps - grad)) / sqrt(length(x))"""
    R = _R(x)
    xeps = x + R(1)
    _, grad_eps = value_and_gradient(f, A @ xeps)
    return R(_norm(A.T @ (grad_eps - grad_f_Ax)) / R(math.sqrt(x.size)))


def backtrack_stepsize_A(gamma, f, A, g, x, f_Ax, At_grad_f_Ax, y, z, g_z, res, Az, grad_f_Az, *, alpha=1.0,
                         minimum_gamma=1e-7, reduce_gamma=0.5):
    """fb_tools.jl:24-63 with a linear map A (Az = A*z is recomputed, :43,:52)."""
    R = _R(x)
    gamma, alpha, minimum_gamma, reduce_gamma = R(gamma), R(alpha), R(minimum_gamma), R(reduce_gamma)
    eps = R(np.finfo(R).eps)
    f_Az_upp = f_model(f_Ax, At_grad_f_Ax, res, alpha / gamma)
    Az[...] = A @ z
    f_Az, grad_tmp = value_and_gradient(f, Az)
    tol = R(10) * eps * (R(1) + abs(f_Az))
    while f_Az > f_Az_upp + tol and gamma >= minimum_gamma:
        gamma = R(gamma * reduce_gamma)
        y[...] = x - gamma * At_grad_f_Ax
        z_new, g_z = prox(g, y, gamma)
        z[...] = z_new
        res[...] = x - z
        f_Az_upp = f_model(f_Ax, At_grad_f_Ax, res, alpha / gamma)
        Az[...] = A @ z
        f_Az, grad_tmp = value_and_gradient(f, Az)
        tol = R(10) * eps * (R(1) + abs(f_Az))
    if grad_f_Az is not None:
        grad_f_Az[...] = grad_tmp
    return gamma, g_z, f_Az, f_Az_upp


class PANOCIteration:
    """panoc.jl:39-52 (options), :87-112 (init), :138-255 (step).  directions: ("lbfgs", M) or None
    (NoAcceleration)."""

    def __init__(self, *, f=None, A, g=None, x0, alpha=0.95, beta=0.5, Lf=None, gamma=None, adaptive=None,
                 minimum_gamma=1e-7, max_backtracks=20, directions=("lbfgs", 5)):
        R = _R(x0)
        self.f = f if f is not None else Zero()
        self.A = np.asarray(A)
        self.g = g if g is not None else Zero()
        self.x0 = x0
        self.alpha, self.beta = R(alpha), R(beta)
        self.Lf = Lf
        self.gamma = gamma if gamma is not None else (None if Lf is None else self.alpha / R(Lf))  # :47
        self.adaptive = (self.gamma is None) if adaptive is None else adaptive  # :48
        self.minimum_gamma = R(minimum_gamma)
        self.max_backtracks = max_backtracks
        self.directions = directions

    def _f_model(self, s):  # :84-85
        return f_model(s.f_Ax, s.At_grad_f_Ax, s.res, self.alpha / s.gamma)

    def init(self):
        R = _R(self.x0)
        A = self.A
        x = self.x0.copy()  # :88
        Ax = A @ x  # :89
        f_Ax, grad_f_Ax = value_and_gradient(self.f, Ax)  # :90
        if self.gamma is None:  # :91-94
            gamma = R(self.alpha / lower_bound_smoothness_constant_A(self.f, A, x, grad_f_Ax))
        else:
            gamma = R(self.gamma)
        At_grad = A.T @ grad_f_Ax  # :95
        y = x - gamma * At_grad  # :96
        z, g_z = prox(self.g, y, gamma)  # :97
        H = LBFGSOperator(self.directions[1], x) if self.directions else None  # :109
        e = lambda v: np.empty_like(v)
        return _State(x=x, Ax=Ax, f_Ax=R(f_Ax), grad_f_Ax=np.array(grad_f_Ax, copy=True), At_grad_f_Ax=At_grad, gamma=gamma,
                      y=y, z=z, g_z=g_z, res=x - z, H=H, tau=R(0), x_prev=e(x), res_prev=e(x), d=e(x), Ad=e(Ax),
                      x_d=e(x), Ax_d=e(Ax), f_Ax_d=R(0), grad_f_Ax_d=e(Ax), At_grad_f_Ax_d=e(x), z_curr=e(x), Az=e(Ax),
                      grad_f_Az=e(Ax), At_grad_f_Az=e(x))

    def step(self, s):
        R = _R(s.x)
        A = self.A
        inf = R(np.inf)
        f_Az, a, b, c = inf, inf, inf, inf  # :139
        if self.adaptive:  # :141-161
            gamma_prev = s.gamma
            s.gamma, s.g_z, f_Az, f_Az_upp = backtrack_stepsize_A(
                s.gamma, self.f, A, self.g, s.x, s.f_Ax, s.At_grad_f_Ax, s.y, s.z, s.g_z, s.res, s.Az, s.grad_f_Az,
                alpha=self.alpha, minimum_gamma=self.minimum_gamma)
            if s.gamma != gamma_prev and s.H is not None:
                s.H.reset()
        else:
            f_Az_upp = self._f_model(s)  # :163
        FBE_x = R(f_Az_upp + s.g_z)  # :167
        if s.H is not None:  # :170 set_next_direction! :114-117
            s.H.mul(s.d, s.res)
            s.d *= R(-1)
        else:
            s.d[...] = -s.res
        s.x_prev[...] = s.x  # :173-174
        s.res_prev[...] = s.res
        s.tau = R(1)  # :177
        s.Ad[...] = A @ s.d  # :178
        s.x_d[...] = s.x + s.d  # :180
        s.Ax_d[...] = s.Ax + s.Ad
        s.f_Ax_d, g_d = value_and_gradient(self.f, s.Ax_d)  # :182
        s.grad_f_Ax_d[...] = g_d
        s.At_grad_f_Ax_d[...] = A.T @ s.grad_f_Ax_d  # :184
        s.x[...] = s.x_d  # :186-191
        s.Ax[...] = s.Ax_d
        s.grad_f_Ax[...] = s.grad_f_Ax_d
        s.At_grad_f_Ax[...] = s.At_grad_f_Ax_d
        s.z_curr[...] = s.z
        s.f_Ax = s.f_Ax_d
        sigma = R(self.beta * (R(0.5) / s.gamma) * (R(1) - self.alpha))  # :193
        tol = R(10) * R(np.finfo(R).eps) * (R(1) + abs(FBE_x))  # :194
        threshold = R(FBE_x - sigma * _norm(s.res) ** 2 + tol)  # :195
        s.y[...] = s.x - s.gamma * s.At_grad_f_Ax  # :197
        z_new, s.g_z = prox(self.g, s.y, s.gamma)  # :198
        s.z[...] = z_new
        s.res[...] = s.x - s.z  # :199
        FBE_x_new = R(self._f_model(s) + s.g_z)  # :200
        quad = getattr(self.f, "is_generalized_quadratic", False)
        for k in range(1, self.max_backtracks + 1):  # :202-250
            if FBE_x_new <= threshold:
                break
            if np.isinf(f_Az):  # :207-209
                s.Az[...] = A @ s.z_curr
            s.tau = R(0) if k >= self.max_backtracks else R(s.tau / R(2))  # :211
            s.x[...] = s.tau * s.x_d + (R(1) - s.tau) * s.z_curr  # :212
            s.Ax[...] = s.tau * s.Ax_d + (R(1) - s.tau) * s.Az  # :213
            if quad:  # :215-237
                if np.isinf(f_Az):
                    f_Az, g_Az = value_and_gradient(self.f, s.Az)
                    s.grad_f_Az[...] = g_Az
                if np.isinf(c):
                    s.At_grad_f_Az[...] = A.T @ s.grad_f_Az
                    c = f_Az
                    b = R(_dot(s.Ax_d, s.grad_f_Az) - _dot(s.Az, s.grad_f_Az))
                    a = R(s.f_Ax_d - b - c)
                s.f_Ax = R(a * s.tau**2 + b * s.tau + c)
                s.grad_f_Ax[...] = s.tau * s.grad_f_Ax_d + (R(1) - s.tau) * s.grad_f_Az
                s.At_grad_f_Ax[...] = s.tau * s.At_grad_f_Ax_d + (R(1) - s.tau) * s.At_grad_f_Az
            else:  # :238-244
                s.f_Ax, g_x = value_and_gradient(self.f, s.Ax)
                s.grad_f_Ax[...] = g_x
                s.At_grad_f_Ax[...] = A.T @ s.grad_f_Ax
            s.y[...] = s.x - s.gamma * s.At_grad_f_Ax  # :246
            z_new, s.g_z = prox(self.g, s.y, s.gamma)  # :247
            s.z[...] = z_new
            s.res[...] = s.x - s.z  # :248
            FBE_x_new = R(self._f_model(s) + s.g_z)  # :249
        if s.H is not None:  # :252 update_direction_state! :122-126
            s.x_prev[...] = s.x - s.x_prev
            s.res_prev[...] = s.res - s.res_prev
            s.H.update(s.x_prev, s.res_prev)
        return s

    def __iter__(self):
        s = self.init()
        yield s
        while True:
            yield self.step(s)


def panoc(*, maxit=1_000, tol=1e-8, **kw):
    """PANOC(; maxit, tol)(; kwargs...)  panoc.jl:297-315; stop :256-257; solution :258 (state.z)."""
    return run(PANOCIteration(**kw), maxit=maxit, tol=tol)


# --------------------------------------------------------------------------------------
# "next" rows: ZeroFPR                          src/algorithms/zerofpr.jl:39-225
# --------------------------------------------------------------------------------------


class ZeroFPRIteration:
    """zerofpr.jl:39-52 (options), :85-111 (init), :142-220 (step)."""

    def __init__(self, *, f=None, A, g=None, x0, alpha=0.95, beta=0.5, Lf=None, gamma=None, adaptive=None,
                 minimum_gamma=1e-7, max_backtracks=20, directions=("lbfgs", 5)):
        R = _R(x0)
        self.f = f if f is not None else Zero()
        self.A = np.asarray(A)
        self.g = g if g is not None else Zero()
        self.x0 = x0
        self.alpha, self.beta = R(alpha), R(beta)
        self.gamma = gamma if gamma is not None else (None if Lf is None else self.alpha / R(Lf))
        self.adaptive = (self.gamma is None) if adaptive is None else adaptive
        self.minimum_gamma = R(minimum_gamma)
        self.max_backtracks = max_backtracks
        self.directions = directions

    def _f_model(self, s):  # :82-83
        return f_model(s.f_Ax, s.At_grad_f_Ax, s.res, self.alpha / s.gamma)

    def init(self):
        R = _R(self.x0)
        A = self.A
        x = self.x0.copy()
        Ax = A @ x
        f_Ax, grad_f_Ax = value_and_gradient(self.f, Ax)
        gamma = R(self.alpha / lower_bound_smoothness_constant_A(self.f, A, x, grad_f_Ax)) if self.gamma is None else R(self.gamma)
        At_grad = A.T @ grad_f_Ax
        y = x - gamma * At_grad
        xbar, g_xbar = prox(self.g, y, gamma)
        H = LBFGSOperator(self.directions[1], x) if self.directions else None
        e = lambda v: np.empty_like(v)
        return _State(x=x, Ax=Ax, f_Ax=R(f_Ax), grad_f_Ax=np.array(grad_f_Ax, copy=True), At_grad_f_Ax=At_grad, gamma=gamma,
                      y=y, xbar=xbar, g_xbar=g_xbar, res=x - xbar, H=H, tau=R(0), Axbar=e(Ax), grad_f_Axbar=e(Ax),
                      At_grad_f_Axbar=e(x), xbarbar=e(x), res_xbar=e(x), xbar_prev=e(x), res_xbar_prev=e(x),
                      is_prev_set=False, d=e(x), Ad=e(Ax))

    def step(self, s):
        R = _R(s.x)
        A = self.A
        if self.adaptive:  # :143-164
            gamma_prev = s.gamma
            s.gamma, s.g_xbar, f_Axbar, f_Axbar_upp = backtrack_stepsize_A(
                s.gamma, self.f, A, self.g, s.x, s.f_Ax, s.At_grad_f_Ax, s.y, s.xbar, s.g_xbar, s.res, s.Axbar,
                s.grad_f_Axbar, alpha=self.alpha, minimum_gamma=self.minimum_gamma)
            if s.gamma != gamma_prev and s.H is not None:
                s.H.reset()
        else:  # :165-170
            s.Axbar[...] = A @ s.xbar
            f_Axbar, g_ = value_and_gradient(self.f, s.Axbar)
            s.grad_f_Axbar[...] = g_
            f_Axbar_upp = self._f_model(s)
        FBE_x = R(f_Axbar_upp + s.g_xbar)  # :173
        s.At_grad_f_Axbar[...] = A.T @ s.grad_f_Axbar  # :176
        s.y[...] = s.xbar - s.gamma * s.At_grad_f_Axbar
        xbb, _ = prox(self.g, s.y, s.gamma)
        s.xbarbar[...] = xbb
        s.res_xbar[...] = s.xbar - s.xbarbar  # :179
        if s.is_prev_set and s.H is not None:  # :181-183, :123-126
            s.xbar_prev[...] = s.xbar - s.xbar_prev
            s.res_xbar_prev[...] = s.res_xbar - s.res_xbar_prev
            s.H.update(s.xbar_prev, s.res_xbar_prev)
        s.xbar_prev[...] = s.xbar  # :185-187
        s.res_xbar_prev[...] = s.res_xbar
        s.is_prev_set = True
        if s.H is not None:  # :189, :113-116
            s.H.mul(s.d, s.res_xbar)
            s.d *= R(-1)
        else:
            s.d[...] = -s.res
        s.tau = R(1)  # :192
        s.Ad[...] = A @ s.d
        sigma = R(self.beta * (R(0.5) / s.gamma) * (R(1) - self.alpha))  # :195
        tol = R(10) * R(np.finfo(R).eps) * (R(1) + abs(FBE_x))
        threshold = R(FBE_x - sigma * _norm(s.res) ** 2 + tol)  # :197
        for k in range(1, self.max_backtracks + 1):  # :199-217
            s.x[...] = s.xbar_prev + s.tau * s.d
            s.Ax[...] = s.Axbar + s.tau * s.Ad
            s.f_Ax, g_ = value_and_gradient(self.f, s.Ax)
            s.grad_f_Ax[...] = g_
            s.At_grad_f_Ax[...] = A.T @ s.grad_f_Ax
            s.y[...] = s.x - s.gamma * s.At_grad_f_Ax
            xb, s.g_xbar = prox(self.g, s.y, s.gamma)
            s.xbar[...] = xb
            s.res[...] = s.x - s.xbar
            FBE_x = R(self._f_model(s) + s.g_xbar)
            if FBE_x <= threshold:
                break
            s.tau = R(0) if k >= self.max_backtracks - 1 else R(s.tau / R(2))
        return s

    def __iter__(self):
        s = self.init()
        yield s
        while True:
            yield self.step(s)


def zerofpr(*, maxit=1_000, tol=1e-8, **kw):
    """ZeroFPR(; maxit, tol)(; kwargs...); stop zerofpr.jl:222-223 ; solution :224 (state.xbar)."""
    it = ZeroFPRIteration(**kw)
    R = _R(it.x0)
    for k, s in enumerate(it, start=1):
        if k >= maxit or _norm_inf(s.res) / s.gamma <= R(tol):
            return s.xbar, k


# --------------------------------------------------------------------------------------
# "next" rows: PANOCplus                        src/algorithms/panocplus.jl:39-250
# --------------------------------------------------------------------------------------


class PANOCplusIteration:
    """panocplus.jl:39-52 (options), :85-128 (init), :168-240 (step)."""

    def __init__(self, *, f=None, A, g=None, x0, alpha=0.95, beta=0.5, Lf=None, gamma=None, adaptive=None,
                 minimum_gamma=1e-7, max_backtracks=20, directions=("lbfgs", 5)):
        R = _R(x0)
        self.f = f if f is not None else Zero()
        self.A = np.asarray(A)
        self.g = g if g is not None else Zero()
        self.x0 = x0
        self.alpha, self.beta = R(alpha), R(beta)
        self.gamma = gamma if gamma is not None else (None if Lf is None else self.alpha / R(Lf))
        self.adaptive = (self.gamma is None) if adaptive is None else adaptive
        self.minimum_gamma = R(minimum_gamma)
        self.max_backtracks = max_backtracks
        self.directions = directions

    def _f_model(self, s):
        return f_model(s.f_Ax, s.At_grad_f_Ax, s.res, self.alpha / s.gamma)

    def init(self):
        R = _R(self.x0)
        A = self.A
        x = self.x0.copy()
        Ax = A @ x
        f_Ax, grad_f_Ax = value_and_gradient(self.f, Ax)
        gamma = R(self.alpha / lower_bound_smoothness_constant_A(self.f, A, x, grad_f_Ax)) if self.gamma is None else R(self.gamma)
        At_grad = A.T @ grad_f_Ax
        y = x - gamma * At_grad
        z, g_z = prox(self.g, y, gamma)
        H = LBFGSOperator(self.directions[1], x) if self.directions else None
        e = lambda v: np.empty_like(v)
        s = _State(x=x, Ax=Ax, f_Ax=R(f_Ax), grad_f_Ax=np.array(grad_f_Ax, copy=True), At_grad_f_Ax=At_grad, gamma=gamma, y=y,
                   z=z, g_z=g_z, res=x - z, H=H, tau=R(0), x_prev=e(x), res_prev=e(x), d=e(x), Az=e(Ax), grad_f_Az=e(Ax),
                   At_grad_f_Az=e(x))
        if self.gamma is None or self.adaptive:  # :105-121
            s.gamma, s.g_z, _, _ = backtrack_stepsize_A(
                s.gamma, self.f, A, self.g, s.x, s.f_Ax, s.At_grad_f_Ax, s.y, s.z, s.g_z, s.res, s.Az, s.grad_f_Az,
                alpha=self.alpha, minimum_gamma=self.minimum_gamma)
        else:  # :122-126
            s.Az[...] = A @ s.z
            _, g_ = value_and_gradient(self.f, s.Az)
            s.grad_f_Az[...] = g_
        s.At_grad_f_Az[...] = A.T @ s.grad_f_Az  # :127
        return s

    def step(self, s):
        R = _R(s.x)
        A = self.A
        s.x_prev[...] = s.x  # :170-171
        s.res_prev[...] = s.res
        FBE_x = R(self._f_model(s) + s.g_z)  # :174
        sigma = R(self.beta * (R(0.5) / s.gamma) * (R(1) - self.alpha))
        tol = R(10) * R(np.finfo(R).eps) * (R(1) + abs(FBE_x))
        threshold = R(FBE_x - sigma * _norm(s.res) ** 2 + tol)  # :178
        tau_backtracks = 0
        can_update_direction = True
        while True:  # :183-235
            if can_update_direction:
                if s.H is not None:  # :130-138
                    s.H.mul(s.d, s.res_prev)
                    s.d *= R(-1)
                else:
                    s.d[...] = -s.res_prev
                s.tau = R(1)
                s.x[...] = s.x_prev + s.d
                tau_backtracks = 0
            else:
                s.x[...] = (R(1) - s.tau) * (s.x_prev - s.res_prev) + s.tau * (s.x_prev + s.d)
                tau_backtracks += 1
            s.Ax[...] = A @ s.x  # :199
            s.f_Ax, g_ = value_and_gradient(self.f, s.Ax)
            s.grad_f_Ax[...] = g_
            s.At_grad_f_Ax[...] = A.T @ s.grad_f_Ax
            s.y[...] = s.x - s.gamma * s.At_grad_f_Ax  # :204
            zz, s.g_z = prox(self.g, s.y, s.gamma)
            s.z[...] = zz
            s.res[...] = s.x - s.z
            f_Az_upp = self._f_model(s)  # :208
            s.Az[...] = A @ s.z  # :210
            f_Az, g_ = value_and_gradient(self.f, s.Az)
            s.grad_f_Az[...] = g_
            if self.gamma is None or self.adaptive:  # :213-224
                tol2 = R(10) * R(np.finfo(R).eps) * (R(1) + abs(f_Az))
                if f_Az > f_Az_upp + tol2 and s.gamma >= self.minimum_gamma:
                    s.gamma = R(s.gamma * R(0.5))
                    can_update_direction = True
                    if s.H is not None:
                        s.H.reset()
                    continue
            s.At_grad_f_Az[...] = A.T @ s.grad_f_Az  # :225
            FBE_x_new = R(f_Az_upp + s.g_z)  # :227
            if FBE_x_new <= threshold or tau_backtracks >= self.max_backtracks:
                break
            s.tau = R(0) if tau_backtracks >= self.max_backtracks - 1 else R(s.tau / R(2))  # :231
            can_update_direction = False
        if s.H is not None:  # :237, :140-148
            s.x_prev[...] = s.x - s.x_prev
            s.res_prev[...] = s.res - s.res_prev
            s.H.update(s.x_prev, s.res_prev)
        return s

    def __iter__(self):
        s = self.init()
        yield s
        while True:
            yield self.step(s)


def panocplus(*, maxit=1_000, tol=1e-8, **kw):
    """PANOCplus(; maxit, tol)(; kwargs...); stop panocplus.jl:243-244:
    norm(res / gamma - At_grad_f_Ax + At_grad_f_Az, Inf) <= tol ; solution :245 (state.z)."""
    it = PANOCplusIteration(**kw)
    R = _R(it.x0)
    for k, s in enumerate(it, start=1):
        if k >= maxit or _norm_inf(s.res / s.gamma - s.At_grad_f_Ax + s.At_grad_f_Az) <= R(tol):
            return s.z, k
