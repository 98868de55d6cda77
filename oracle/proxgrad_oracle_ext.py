"""CPU oracle, part 2: the remaining proximal-gradient / splitting iterations of the reference.

*** TEST INFRASTRUCTURE ONLY *** (same rule as proxgrad_oracle.py: only tests/, smoke() and bench.py's CPU leg may
import it).  numpy restatements -- same operation order, working precision R = dtype of x0 -- of

  * SFISTAIteration       src/algorithms/sfista.jl:56-92, stop :94-108, solution :111
  * DavisYinIteration     src/algorithms/davis_yin.jl:41-84, stop :86-87, solution :88
  * LiLinIteration        src/algorithms/li_lin.jl:40-144, stop :146-147, solution :148
  * DRLSIteration         src/algorithms/drls.jl:26-40 (defaults), :65-197 (iterate), stop :199-200, solution :201
  * AFBAIteration         src/algorithms/primal_dual.jl:83-112 (options), :176-209 (iterate), :334-416 (default steps);
                          VuCondat = AFBA(theta=2) :131, ChambollePock = AFBA(theta=2, f=Zero, l=IndZero) :151-152

Pins (tests/test_oracle_golden.py): test_lasso_small.jl:216-231 (DRLS), :233-272 (AFBA), :274-283 (SFISTA),
test_lasso_small_strongly_convex.jl (SFISTA, DRLS), test_elasticnet.jl:31-56 (DavisYin), :58-120 (AFBA variants),
test_nonconvex_qp.jl:58-66 (LiLin), test_equivalence.jl:14-49 (DR == DRLS without acceleration).
The operators SqrNormL2 / Translate / conjugates live in the un-vendored ProximalOperators.jl (compat 0.15): their
published closed forms are restated here.
"""
from __future__ import annotations

import math

import numpy as np

from .proxgrad_oracle import (LBFGSOperator, Zero, _dot, _norm, _norm_inf, _R, _State, fixed_nesterov_sequence, prox,
                              simple_nesterov_sequence, value_and_gradient)

# --------------------------------------------------------------------------------------
# operators of the pins
# --------------------------------------------------------------------------------------


class SqrNormL2:
    """f(x) = lam/2 ||x||^2 (ProximalOperators.SqrNormL2(lam)): prox = x / (1 + lam gamma); gradient lam x."""

    def __init__(self, lam=1.0):
        self.lam = lam

    def prox(self, x, gamma):
        R = _R(x)
        y = (x / (R(1) + R(self.lam) * R(gamma))).astype(x.dtype)
        return y, self(y)

    def value_and_gradient(self, x):
        return self(x), (_R(x)(self.lam) * x).astype(x.dtype)

    def __call__(self, x):
        R = _R(x)
        return R(R(self.lam) / R(2) * _norm(x) ** 2)


class SqrDistance:
    """f(u) = lam/2 ||u - b||^2 (ProximalOperators: Translate(SqrNormL2(lam), -b), test_elasticnet.jl:24 and the
    `f_prox` of test_lasso_small.jl:38): prox = (u + lam gamma b) / (1 + lam gamma)."""

    def __init__(self, b, lam=1.0):
        self.b, self.lam = b, lam

    def prox(self, x, gamma):
        R = _R(x)
        lg = R(self.lam) * R(gamma)
        y = ((x + lg * self.b.astype(x.dtype)) / (R(1) + lg)).astype(x.dtype)
        return y, self(y)

    def value_and_gradient(self, x):
        R = _R(x)
        d = x - self.b.astype(x.dtype)
        return R(R(self.lam) / R(2) * _norm(d) ** 2), (R(self.lam) * d).astype(x.dtype)

    def __call__(self, x):
        return self.value_and_gradient(x)[0]


class Conjugate:
    """ProximalCore.convex_conjugate(f): prox through Moreau's identity
    prox_{gamma f*}(x) = x - gamma prox_{f / gamma}(x / gamma); the value returned is f*(y) = <y, p> - f(p) with
    p = prox_{f/gamma}(x/gamma) (ProximalOperators' Conjugate)."""

    def __init__(self, f):
        self.f = f

    def prox(self, x, gamma):
        R = _R(x)
        p, fp = prox(self.f, (x / R(gamma)).astype(x.dtype), R(1) / R(gamma))
        y = (x - R(gamma) * p).astype(x.dtype)
        return y, R(_dot(y, p) - fp)


class IndZero:
    """ProximalCore.IndZero: indicator of {0}; its conjugate is Zero (value_and_gradient = (0, 0))."""


def convex_conjugate(f):
    if isinstance(f, IndZero):
        return Zero()
    if isinstance(f, Conjugate):
        return f.f
    if isinstance(f, SqrNormL2) and f.lam > 0:
        return SqrNormL2(1.0 / f.lam)  # exact conjugate of lam/2 ||.||^2
    return Conjugate(f)


class Linear:
    """f(x) = <c, x> (ProximalOperators.Linear; the closure `x -> dot(c, x)` of test_linear_programs.jl:107):
    gradient c, prox = x - gamma c."""

    def __init__(self, c):
        self.c = c

    def value_and_gradient(self, x):
        return _R(x)(_dot(self.c.astype(x.dtype), x)), self.c.astype(x.dtype).copy()

    def prox(self, x, gamma):
        y = (x - _R(x)(gamma) * self.c.astype(x.dtype)).astype(x.dtype)
        return y, _R(x)(_dot(self.c.astype(x.dtype), y))

    def __call__(self, x):
        return self.value_and_gradient(x)[0]


class IndNonnegative:
    """indicator of {x >= 0} (ProximalOperators.IndNonnegative): prox = max.(0, x)"""

    def prox(self, x, gamma):
        return np.maximum(x, _R(x)(0)), _R(x)(0)

    def __call__(self, x):
        return _R(x)(0) if np.all(x >= 0) else _R(x)(np.inf)


class IndPoint:
    """indicator of {b} (ProximalOperators.IndPoint): prox = b"""

    def __init__(self, b):
        self.b = b

    def prox(self, x, gamma):
        return self.b.astype(x.dtype).copy(), _R(x)(0)


class IndAffine:
    """indicator of {x : A x = b} (ProximalOperators.IndAffine, A with full row rank):
    prox = x - A' (A A')^{-1} (A x - b)"""

    def __init__(self, A, b):
        self.A, self.b = np.asarray(A), np.asarray(b)

    def prox(self, x, gamma):
        A = self.A.astype(x.dtype)
        res = A @ x - self.b.astype(x.dtype)
        return (x - A.T @ np.linalg.solve(A @ A.T, res)).astype(x.dtype), _R(x)(0)


class SlicedSeparableSum:
    """ProximalOperators.SlicedSeparableSum for contiguous index ranges: h(y) = sum_k h_k(y[lo_k:hi_k])"""

    def __init__(self, fs, ranges):
        self.fs, self.ranges = fs, ranges

    def prox(self, x, gamma):
        y = np.empty_like(x)
        v = _R(x)(0)
        for f, (lo, hi) in zip(self.fs, self.ranges):
            y[lo:hi], vk = prox(f, x[lo:hi], gamma)
            v = _R(x)(v + vk)
        return y, v


def _isapprox(a, b, R):
    """Julia isapprox for reals: |a - b| <= sqrt(eps(R)) * max(|a|, |b|)"""
    return abs(a - b) <= math.sqrt(np.finfo(R).eps) * max(abs(a), abs(b))


# --------------------------------------------------------------------------------------
# SFISTA                                          src/algorithms/sfista.jl
# --------------------------------------------------------------------------------------


class SFISTAIteration:
    """sfista.jl:37-43 (options x0, f, g, Lf, mf) and Base.iterate :56-92."""

    def __init__(self, *, x0, f=None, g=None, Lf, mf=0.0):
        self.x0, self.f, self.g = x0, f if f is not None else Zero(), g if g is not None else Zero()
        R = _R(x0)
        self.Lf, self.mf = R(Lf), R(mf)

    def __iter__(self):
        R = _R(self.x0)
        lam = R(1) / self.Lf  # :58  state = SFISTAState(lambda = 1 / Lf, yPrev = copy(x0))
        s = _State(lam=lam, yPrev=self.x0.copy(), y=np.zeros_like(self.x0), xPrev=self.x0.copy(), x=np.zeros_like(self.x0),
                   xt=np.zeros_like(self.x0), tau=R(1), a=R(0), APrev=R(1), A=R(0), gradf_xt=np.zeros_like(self.x0))
        while True:
            s.tau = R(s.lam * (R(1) + self.mf * s.APrev))  # :61
            s.a = R((s.tau + R(np.sqrt(R(s.tau * s.tau + R(4) * s.tau * s.APrev)))) / R(2))  # :62
            s.A = R(s.APrev + s.a)  # :63
            s.xt[...] = R(s.APrev / s.A) * s.yPrev + R(s.a / s.A) * s.xPrev  # :64
            _, g = value_and_gradient(self.f, s.xt)  # :65
            s.gradf_xt[...] = g
            lam2 = R(s.lam / (R(1) + s.lam * self.mf))  # :67
            y, _ = prox(self.g, (s.xt - lam2 * s.gradf_xt).astype(s.xt.dtype), lam2)  # :69
            s.y[...] = y
            c = R(s.a / (R(1) + s.A * self.mf))
            s.x[...] = s.xPrev + c * ((s.y - s.xt) / s.lam + self.mf * (s.y - s.xPrev))  # :70-73
            s.yPrev[...] = s.y  # :75-77
            s.xPrev[...] = s.x
            s.APrev = s.A
            yield s


def sfista_residual(it, s):
    """check_sc, classic termination (sfista.jl:101-106): r = grad f(y) - grad f(xt) + (xt - y) / lam2 ; ||r||."""
    R = _R(it.x0)
    lam2 = R(s.lam / (R(1) + s.lam * it.mf))
    _, gy = value_and_gradient(it.f, s.y)
    return R(_norm(gy - s.gradf_xt + (s.xt - s.y) / lam2))


def sfista(*, maxit=10_000, tol=1e-6, **kw):
    """SFISTA(; maxit, tol)(; kwargs...)  sfista.jl:140-160; stop = res <= tol || res ~ tol (:107); solution y (:111)."""
    it = SFISTAIteration(**kw)
    R = _R(it.x0)
    for k, s in enumerate(it, start=1):
        res = sfista_residual(it, s)
        if k >= maxit or res <= R(tol) or _isapprox(float(res), float(R(tol)), R):
            return s.y, k


# --------------------------------------------------------------------------------------
# Davis-Yin three-operator splitting              src/algorithms/davis_yin.jl
# --------------------------------------------------------------------------------------


class DavisYinIteration:
    """davis_yin.jl:41-50 (options f, g, h, x0, lambda = 1, Lf | gamma = 1 / Lf) and iterate :62-84:
        xg = prox_g(z); z_half = 2 xg - z - gamma grad f(xg); xh = prox_h(z_half); res = xh - xg; z += lambda res"""

    def __init__(self, *, x0, f=None, g=None, h=None, lam=1.0, Lf=None, gamma=None):
        self.x0 = x0
        self.f, self.g, self.h = (o if o is not None else Zero() for o in (f, g, h))
        R = _R(x0)
        if gamma is None:
            if Lf is None:
                raise ValueError("You must specify either Lf or gamma")  # :48-49
            gamma = R(1) / R(Lf)
        self.gamma, self.lam = R(gamma), R(lam)

    def __iter__(self):
        s = _State(z=self.x0.copy())
        while True:
            s.xg, _ = prox(self.g, s.z, self.gamma)
            _, s.grad_f_xg = value_and_gradient(self.f, s.xg)
            s.z_half = (2 * s.xg - s.z - self.gamma * s.grad_f_xg).astype(s.z.dtype)
            s.xh, _ = prox(self.h, s.z_half, self.gamma)
            s.res = s.xh - s.xg
            s.z = (s.z + self.lam * s.res).astype(s.z.dtype)
            yield s


def davis_yin(*, maxit=10_000, tol=1e-8, **kw):
    """DavisYin(; maxit, tol)(; kwargs...)  davis_yin.jl:121-139; stop norm(res, Inf) <= tol (:86-87); solution xh."""
    it = DavisYinIteration(**kw)
    R = _R(it.x0)
    for k, s in enumerate(it, start=1):
        if k >= maxit or _norm_inf(s.res) <= R(tol):
            return s.xh, k


# --------------------------------------------------------------------------------------
# Li-Lin nonconvex accelerated proximal gradient   src/algorithms/li_lin.jl
# --------------------------------------------------------------------------------------


class LiLinIteration:
    """li_lin.jl:40-49 (options f, g, x0, Lf | gamma, adaptive, delta = 1e-3, eta = 0.8), init :69-97, step :99-144.

    NOTE li_lin.jl:108 evaluates ``value_and_gradient(iter.f, x)`` with an unbound name ``x`` -- the monitor branch
    (``Fz`` above the moving average) raises UndefVarError in the reference, so nothing it contains is executable there;
    on the reference's own pins the branch is never taken.  The restatement follows Algorithm 2 of Li & Lin (2015) in
    that branch: the gradient at ``state.x``, and for the case x+ = v the extrapolation
    y = v + (t/t+)(z - v) + ((t - 1)/t+)(v - x)  (li_lin.jl:120-122 writes ``z +`` where the paper has ``v +``; taken
    literally that line makes the iteration diverge on a convex LASSO as soon as the branch fires, for every step size)."""

    def __init__(self, *, x0, f=None, g=None, Lf=None, gamma=None, adaptive=False, delta=1e-3, eta=0.8):
        self.x0, self.f, self.g = x0, f if f is not None else Zero(), g if g is not None else Zero()
        R = _R(x0)
        if gamma is None and Lf is not None:
            gamma = R(1) / R(Lf)
        self.gamma = None if gamma is None else R(gamma)
        self.adaptive, self.delta, self.eta = adaptive, R(delta), R(eta)
        self.monitor_branch_taken = 0

    def __iter__(self):
        R = _R(self.x0)
        y = self.x0.copy()
        f_y, grad_f_y = value_and_gradient(self.f, y)
        y_forward = (y - self.gamma * grad_f_y).astype(y.dtype)
        z, g_z = prox(self.g, y_forward, self.gamma)
        Fy = R(f_y + self.g(y))
        assert np.isfinite(Fy), "initial point must be feasible"
        s = _State(x=self.x0.copy(), y=y, f_y=f_y, grad_f_y=grad_f_y, gamma=self.gamma, y_forward=y_forward, z=z, g_z=g_z,
                   res=y - z, theta=R(1), F_average=Fy, q=R(1))
        yield s
        while True:
            Fz = R(self.f(s.z) + s.g_z)  # :103
            theta1 = R((R(1) + R(np.sqrt(R(R(1) + R(4) * s.theta * s.theta)))) / R(2))  # :104
            v = Fv = None
            if Fz <= s.F_average - self.delta * _norm(s.res) ** 2:  # :106
                case = 1
            else:
                self.monitor_branch_taken += 1
                _, grad_f_x = value_and_gradient(self.f, s.x)  # :108 (see the class note)
                v, g_v = prox(self.g, (s.x - s.gamma * grad_f_x).astype(s.x.dtype), s.gamma)
                Fv = R(self.f(v) + g_v)
                case = 1 if Fz <= Fv else 2
            if case == 1:
                s.y = (s.z + R((s.theta - R(1)) / theta1) * (s.z - s.x)).astype(s.z.dtype)  # :116
                s.x, s.z = s.z, s.x
                Fx = Fz
            else:
                s.y = (v + R(s.theta / theta1) * (s.z - v) + R((s.theta - R(1)) / theta1) * (v - s.x)).astype(s.z.dtype)  # see the class note
                s.x = v
                Fx = Fv
            s.f_y, g = value_and_gradient(self.f, s.y)  # :128
            s.grad_f_y = g
            s.y_forward = (s.y - s.gamma * s.grad_f_y).astype(s.y.dtype)
            s.z, s.g_z = prox(self.g, s.y_forward, s.gamma)
            s.res = s.y - s.z
            s.theta = theta1
            q1 = R(self.eta * s.q + R(1))  # :139-141
            s.F_average = R((self.eta * s.q * s.F_average + Fx) / q1)
            s.q = q1
            yield s


def li_lin(*, maxit=10_000, tol=1e-8, **kw):
    """LiLin(; maxit, tol)(; kwargs...)  li_lin.jl:176-194; stop norm(res, Inf) / gamma <= tol; solution z."""
    it = LiLinIteration(**kw)
    R = _R(it.x0)
    for k, s in enumerate(it, start=1):
        if k >= maxit or _norm_inf(s.res) / s.gamma <= R(tol):
            return s.z, k


# --------------------------------------------------------------------------------------
# Anderson and Broyden accelerations               src/accel/anderson.jl, src/accel/broyden.jl
# --------------------------------------------------------------------------------------


class AndersonAccelerationOperator:
    """anderson.jl:5-62: circular memory of M (s, y) pairs; mul!: d = v + (S - Y) pinv(Y'Y) Y'v."""

    def __init__(self, M, x):
        self.M, self.currmem, self.curridx = M, 0, 0
        self.s_M = [np.zeros_like(x) for _ in range(M)]
        self.y_M = [np.zeros_like(x) for _ in range(M)]

    def update(self, s, y):
        self.curridx = 1 if self.curridx + 1 > self.M else self.curridx + 1  # :29-32
        self.currmem = min(self.currmem + 1, self.M)  # :33-36
        self.s_M[self.curridx - 1][...] = s
        self.y_M[self.curridx - 1][...] = y
        return self

    def reset(self):
        self.currmem = self.curridx = 0

    def __mul__(self, v):
        if self.currmem == 0:
            return v.copy()
        S = np.stack(self.s_M[: self.currmem], axis=1)
        Y = np.stack(self.y_M[: self.currmem], axis=1)
        return (v + (S - Y) @ (np.linalg.pinv(Y.T @ Y) @ (Y.T @ v))).astype(v.dtype)  # :59


class BroydenOperator:
    """broyden.jl:5-43: dense H (identity at start), Powell-damped rank-one update."""

    def __init__(self, x, theta_bar=0.2):
        self.H = np.eye(x.size, dtype=x.dtype)
        self.theta_bar = _R(x)(theta_bar)

    def update(self, s, y):
        R = _R(s)
        Hy = self.H @ y  # :19
        sH = s @ self.H  # :20
        delta = R(_dot(Hy, s) / _norm(s) ** 2)  # :21
        if abs(delta) >= self.theta_bar:  # :22-26
            theta = R(1)
        else:
            sgn = R(1) if delta == 0 else R(np.sign(delta))
            theta = R((R(1) - sgn * self.theta_bar) / (R(1) - delta))
        self.H += np.outer((s - Hy) / R(_dot(s, (R(1) / theta - R(1)) * s + Hy)), sH).astype(s.dtype)  # :27
        return self

    def reset(self):
        self.H[...] = np.eye(self.H.shape[0], dtype=self.H.dtype)

    def __mul__(self, v):
        return (self.H @ v).astype(v.dtype)


# --------------------------------------------------------------------------------------
# Douglas-Rachford line search (DRLS)              src/algorithms/drls.jl
# --------------------------------------------------------------------------------------


def drls_default_gamma(convex, mf, Lf, alpha, lam):
    """drls.jl:12-17"""
    if mf is not None and mf > 0:
        return 1 / (alpha * mf)
    return alpha / Lf if convex else alpha * (2 - lam) / (2 * Lf)


def drls_C(convex, mf, Lf, gamma, lam):
    """drls.jl:19-23"""
    a = gamma * Lf if (mf is None or mf <= 0) else 1 / (gamma * mf)
    m = max(a - lam / 2, 0) if convex else 1
    return lam / ((1 + a) ** 2) * ((2 - lam) / 2 - a * m)


class DRLSIteration:
    """drls.jl:65-80 (options; ``directions`` in {"lbfgs", "anderson" (memory 5), "broyden", "nesterov_fixed", "nesterov_simple", "none"}),
    init :112-134, direction hooks :136-158, step :160-197.  ``f_convex`` / ``f_quadratic`` restate the traits
    ProximalCore.is_convex / is_generalized_quadratic of the operator type (true for LeastSquares and the quadratics
    used by the pins)."""

    def __init__(self, *, x0, f=None, g=None, alpha=0.95, beta=0.5, lam=1.0, mf=None, Lf=None, gamma=None, c=None,
                 dre_sign=None, max_backtracks=20, directions="lbfgs", memory=5, f_convex=True, f_quadratic=True):
        self.x0, self.f, self.g = x0, f if f is not None else Zero(), g if g is not None else Zero()
        R = _R(x0)
        self.alpha, self.beta, self.lam = R(alpha), R(beta), R(lam)
        self.mf, self.Lf = mf, Lf
        if gamma is None:
            gamma = drls_default_gamma(f_convex, mf, Lf, self.alpha, self.lam)
        self.gamma = R(gamma)
        self.c = R(c) if c is not None else R(self.beta * R(drls_C(f_convex, mf, Lf, self.gamma, self.lam)))
        self.dre_sign = dre_sign if dre_sign is not None else (1 if (mf is None or mf <= 0) else -1)
        self.max_backtracks, self.directions, self.memory = max_backtracks, directions, memory
        self.f_quadratic = f_quadratic

    def dre(self, s):
        """DRE :105-111: f(u) + g(v) - <x - u, res> / gamma + ||res||^2 / (2 gamma)"""
        R = _R(self.x0)
        return R(s.f_u + s.g_v - R(_dot(s.x - s.u, s.res)) / self.gamma + R(1) / (R(2) * self.gamma) * _norm(s.res) ** 2)

    def _dr_tail(self, s):
        s.w = (2 * s.u - s.x).astype(s.x.dtype)
        s.v, s.g_v = prox(self.g, s.w, self.gamma)
        s.res = s.u - s.v
        s.xbar = (s.x - self.lam * s.res).astype(s.x.dtype)

    def __iter__(self):
        R = _R(self.x0)
        s = _State(x=self.x0.copy(), gamma=self.gamma, tau=R(0))
        s.u, s.f_u = prox(self.f, s.x, self.gamma)
        self._dr_tail(s)
        s.xbar_prev = s.xbar.copy()
        s.res_prev = np.empty_like(s.x)
        H = seq = None
        if self.directions == "lbfgs":
            H = LBFGSOperator(self.memory, s.x)
        elif self.directions == "broyden":
            H = BroydenOperator(s.x)
        elif self.directions == "anderson":
            H = AndersonAccelerationOperator(self.memory, s.x)
        elif self.directions in ("nesterov_fixed", "nesterov_simple"):
            seq = fixed_nesterov_sequence(R) if self.directions == "nesterov_fixed" else simple_nesterov_sequence(R)
        s.H = H
        yield s
        while True:
            dre_curr = self.dre(s)
            threshold = R(self.dre_sign * dre_curr - self.c / self.gamma * _norm(s.res) ** 2)  # :163
            if H is not None:  # :137-140
                s.d = -(H * s.res)
            elif seq is not None:  # :142-144
                s.d = (next(seq) * (s.xbar - s.xbar_prev) + (s.xbar - s.x)).astype(s.x.dtype)
            else:  # :146-147
                s.d = s.xbar - s.x
            s.x_d = s.x + s.d
            s.xbar_prev, s.xbar = s.xbar, s.xbar_prev  # :168-169
            s.res_prev, s.res = s.res, s.res_prev
            s.tau = R(1)
            s.x = s.x_d.copy()
            s.u, s.f_u = prox(self.f, s.x, self.gamma)  # :174
            self._dr_tail(s)
            if H is not None:  # :151-154
                s.res_prev = s.res - s.res_prev
                H.update(s.d, s.res_prev)
            a = b = c = R(0)
            for k in range(1, self.max_backtracks + 1):  # :183-195
                if self.dre_sign * self.dre(s) <= threshold:
                    break
                s.tau = R(0) if k == self.max_backtracks else R(s.tau / R(2))
                s.x = (s.tau * s.x_d + (R(1) - s.tau) * s.xbar_prev).astype(s.x.dtype)
                if self.f_quadratic:
                    if k == 1:
                        u1 = s.u.copy()
                        u0, c = prox(self.f, s.xbar_prev, self.gamma)
                        b = R(R(_dot(s.xbar_prev - s.x_d, s.xbar_prev - u0)) / self.gamma)
                        a = R(s.f_u - b - c)
                    s.u = (s.tau * u1 + (R(1) - s.tau) * u0).astype(s.x.dtype)
                    s.f_u = R(a * s.tau * s.tau + b * s.tau + c)
                else:
                    s.u, s.f_u = prox(self.f, s.x, self.gamma)
                self._dr_tail(s)
            yield s


def drls(*, maxit=1_000, tol=1e-8, **kw):
    """DRLS(; maxit, tol)(; kwargs...)  drls.jl:235-253; stop norm(res, Inf) / gamma <= tol; solution v."""
    it = DRLSIteration(**kw)
    R = _R(it.x0)
    for k, s in enumerate(it, start=1):
        if k >= maxit or _norm_inf(s.res) / s.gamma <= R(tol):
            return s.v, k


# --------------------------------------------------------------------------------------
# AFBA / Vu-Condat / Chambolle-Pock                src/algorithms/primal_dual.jl
# --------------------------------------------------------------------------------------


def afba_default_stepsizes(nmL, h_is_zero, theta, mu, beta_f, beta_l, R):
    """primal_dual.jl:334-416.  ``nmL`` = opnorm(L).  The rule picks a balance ``alpha`` between the primal and dual
    steps from the relative sizes of ||L||, beta_f, beta_l, then gamma1 = 1 / (beta_f / 2 + c1 ||L|| / alpha),
    gamma2 = 0.99 / (beta_l / 2 + c2 ||L|| alpha) with method-dependent factors c1, c2."""
    theta, mu, beta_f, beta_l, nmL = R(theta), R(mu), R(beta_f), R(beta_l), R(nmL)
    if h_is_zero:
        return R(1.99) / beta_f, R(1)
    par, par2 = R(5), R(100)
    ap = lambda a, b: _isapprox(float(a), float(b), R)

    def balance(n):
        if n > par * max(beta_l, beta_f):
            return R(1)
        if beta_f > par * beta_l:
            return par2 * n / beta_f
        if beta_l > par * beta_f:
            return beta_l / (par2 * n)
        return R(1)

    def steps(n, alpha, c1=R(1), c2=R(1)):
        return R(1) / (beta_f / 2 + c1 * n / alpha), R(0.99) / (beta_l / 2 + c2 * n * alpha)

    if ap(theta, 2):  # Vu-Condat
        return steps(nmL, balance(nmL))
    if ap(theta, 1) and ap(mu, 1):  # SPCA
        alpha = R(1)
        if nmL > par2 * beta_l:
            alpha = R(1)
        elif beta_l > par * beta_f:
            alpha = beta_l / (par2 * nmL)
        g1 = R(1.99) / beta_f if beta_f > 0 else R(1) / (nmL / alpha)
        return g1, R(0.99) / (beta_l / 2 + g1 * nmL * nmL)
    if ap(theta, 0) and ap(mu, 1):  # PPCA
        if ap(beta_f, 0):
            n = R(nmL * R(np.sqrt(R(3))))
            return steps(n, R(1) if n > par * beta_l else beta_l / (par2 * n))
        alpha = balance(nmL)
        xi = 1 + 2 * nmL / (nmL + alpha * beta_f / 2)
        return steps(nmL, alpha, c2=xi)
    if ap(mu, 0):  # SDCA, PDCA
        temp = theta * theta - 3 * theta + 3
        if ap(beta_l, 0):
            n = R(nmL * R(np.sqrt(temp)))
            return steps(n, R(1) if n > par * beta_f else par2 * n / beta_f)
        alpha = balance(nmL)
        eta = 1 + (temp - 1) * alpha * nmL / (alpha * nmL + beta_l / 2)
        return steps(nmL, alpha, c1=eta)
    if ap(theta, 0) and ap(mu, 0.5):  # PPDCA
        alpha = balance(nmL) if (ap(beta_l, 0) or ap(beta_f, 0)) else R(np.sqrt(beta_l / beta_f)) / 2
        return steps(nmL, alpha)
    raise ValueError("this choice of theta and mu is not supported!")


class AFBAIteration:
    """primal_dual.jl:83-112 and Base.iterate :176-209.  ``L`` is a dense matrix, ``None`` meaning the identity (or
    0 * I when h is Zero, :87-91)."""

    def __init__(self, *, x0, y0, f=None, g=None, h=None, l=None, L=None, beta_f=None, beta_l=None, theta=1.0, mu=1.0,
                 lam=1.0, gamma=None):
        R = _R(x0)
        self.x0, self.y0 = x0, y0
        self.f, self.g, self.h = (o if o is not None else Zero() for o in (f, g, h))
        self.l = l if l is not None else IndZero()
        self.L, self.h_is_zero = L, isinstance(self.h, Zero)
        if beta_f is None:
            if not isinstance(self.f, Zero):
                raise ValueError("argument beta_f must be specified together with f")
            beta_f = 0
        if beta_l is None:
            if not isinstance(self.l, IndZero):
                raise ValueError("argument beta_l must be specified together with l")
            beta_l = 0
        self.theta, self.mu, self.lam = R(theta), R(mu), R(lam)
        if gamma is None:
            if self.lam != 1:
                raise ValueError("if lambda != 1, then you need to provide stepsizes manually")
            nmL = 0.0 if self.h_is_zero else (1.0 if L is None else np.linalg.norm(np.asarray(L, np.float64), 2))
            gamma = afba_default_stepsizes(nmL, self.h_is_zero, theta, mu, beta_f, beta_l, R)
        self.gamma = (R(gamma[0]), R(gamma[1]))

    def _L(self, x):
        if self.h_is_zero and self.L is None:
            return np.zeros_like(self.y0)
        return x.copy() if self.L is None else (self.L @ x).astype(x.dtype)

    def _Lt(self, y):
        if self.h_is_zero and self.L is None:
            return np.zeros_like(self.x0)
        return y.copy() if self.L is None else (self.L.T @ y).astype(y.dtype)

    def __iter__(self):
        R = _R(self.x0)
        g1, g2 = self.gamma
        s = _State(x=self.x0.copy(), y=self.y0.copy())
        hc, lc = convex_conjugate(self.h), convex_conjugate(self.l)
        while True:
            _, s.gradf = value_and_gradient(self.f, s.x)  # :180
            t = self._Lt(s.y) + s.gradf  # :182-185
            t = (s.x + (-g1) * t).astype(s.x.dtype)
            s.xbar, _ = prox(self.g, t, g1)  # :186
            _, s.gradl = value_and_gradient(lc, s.y)  # :187
            t = (self.theta * s.xbar + (R(1) - self.theta) * s.x).astype(s.x.dtype)  # :189
            ty = self._L(t) - s.gradl  # :190-193
            ty = (s.y + g2 * ty).astype(s.y.dtype)
            s.ybar, _ = prox(hc, ty, g2)  # :194
            s.FPR_x = s.xbar - s.x  # :196-197
            s.FPR_y = s.ybar - s.y
            ty = (R(self.mu * (R(2) - self.theta) * g1) * s.FPR_y).astype(s.y.dtype)  # :199-201
            s.x = (s.x + self.lam * (s.FPR_x - self._Lt(ty))).astype(s.x.dtype)
            t = (R((R(1) - self.mu) * (R(2) - self.theta) * g2) * s.FPR_x).astype(s.x.dtype)  # :203-205
            s.y = (s.y + self.lam * (s.FPR_y + self._L(t))).astype(s.y.dtype)
            yield s


def afba(*, maxit=10_000, tol=1e-5, **kw):
    """AFBA(; maxit, tol)(; kwargs...)  primal_dual.jl:251-269; stop ||FPR_x||_inf + ||FPR_y||_inf <= tol (:211-212);
    solution (xbar, ybar) (:213)."""
    it = AFBAIteration(**kw)
    R = _R(it.x0)
    for k, s in enumerate(it, start=1):
        if k >= maxit or _norm_inf(s.FPR_x) + _norm_inf(s.FPR_y) <= R(tol):
            return (s.xbar, s.ybar), k


def vu_condat(**kw):
    """VuCondat = AFBA with theta = 2 (primal_dual.jl:131, :297-298)"""
    return afba(theta=2, **kw)


def chambolle_pock(**kw):
    """ChambollePock = AFBA with theta = 2, f = Zero, l = IndZero (primal_dual.jl:151-152, :328-329)"""
    kw.pop("f", None), kw.pop("l", None)
    return afba(theta=2, **kw)
