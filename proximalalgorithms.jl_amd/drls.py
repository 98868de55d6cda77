"""Douglas-Rachford line search (DRLS) -- mirror of src/algorithms/drls.jl (Themelis, Stella, Patrinos 2020).

A Douglas-Rachford step followed by a quasi-Newton / Nesterov direction and a backtracking line search on the
Douglas-Rachford envelope.  f needs ``prox_`` (SeparableQuadratic, SqrNormL2, SquaredDistance, LeastSquares); the
direction memory is a device operator (L-BFGS, Broyden or Anderson); every vector statement is a HIP kernel of the
library (prox, AXPBY, dot, norms).
"""
import numpy as np

from .algorithm import IterativeAlgorithm
from .device import as_hipvector
from .lbfgs import LBFGS
from .operators import Zero, is_convex, is_generalized_quadratic, prox_
from .panoc import NoAcceleration


def drls_default_gamma(f, mf, Lf, alpha, lam):
    """drls.jl:12-17"""
    if mf is not None and mf > 0:
        return 1 / (alpha * mf)
    return alpha / Lf if is_convex(f) else alpha * (2 - lam) / (2 * Lf)


def drls_C(f, mf, Lf, gamma, lam):
    """drls.jl:19-23"""
    a = gamma * Lf if (mf is None or mf <= 0) else 1 / (gamma * mf)
    m = max(a - lam / 2, 0) if is_convex(f) else 1
    return lam / ((1 + a) ** 2) * ((2 - lam) / 2 - a * m)


class DRLSState:
    """drls.jl:84-104"""


class DRLSIteration:
    """drls.jl:65-80 (options) ; init :112-134 ; direction hooks :136-158 ; step :160-197"""

    def __init__(self, *, x0, f=None, g=None, alpha=0.95, beta=0.5, lam=1.0, mf=None, Lf=None, gamma=None, c=None,
                 dre_sign=None, max_backtracks=20, directions=None, **kw):
        if "lambda_" in kw:
            lam = kw.pop("lambda_")
        if kw:
            raise TypeError(f"unexpected keyword arguments {sorted(kw)}")
        self.x0 = as_hipvector(x0)
        R = self.x0.dtype.type
        self.f = f if f is not None else Zero()
        self.g = g if g is not None else Zero()
        self.alpha, self.beta, self.lam = R(alpha), R(beta), R(lam)
        self.mf, self.Lf = mf, Lf
        if gamma is None:
            if Lf is None and not (mf is not None and mf > 0):
                raise ValueError("one of gamma, Lf or mf > 0 must be given")
            gamma = drls_default_gamma(self.f, mf, Lf, self.alpha, self.lam)
        self.gamma = R(gamma)
        if c is None:
            if Lf is None and not (mf is not None and mf > 0):
                raise ValueError("c needs Lf or mf > 0")
            c = self.beta * R(drls_C(self.f, mf, Lf, self.gamma, self.lam))
        self.c = R(c)
        self.dre_sign = int(dre_sign) if dre_sign is not None else (1 if (mf is None or mf <= 0) else -1)
        self.max_backtracks = int(max_backtracks)
        self.directions = directions if directions is not None else LBFGS(5)

    def DRE(self, s):
        """drls.jl:105-111: f(u) + g(v) - <x - u, res> / gamma + ||res||^2 / (2 gamma).  The value (and ||res||^2) of the
        state the line search accepted is kept: the next iteration starts from exactly that state (:161-163), so its
        envelope and threshold cost no further reductions."""
        R = s.x.dtype.type
        dot_product = R(s.x.dot(s.res) - s.u.dot(s.res))
        s.res_sq = R(s.res.norm() ** 2)
        s.dre = R(s.f_u + s.g_v - dot_product / self.gamma + R(1) / (R(2) * self.gamma) * s.res_sq)
        return s.dre

    def _dr_tail(self, s):
        s.w.axpby_(2.0, s.u, -1.0, s.x)
        s.g_v = prox_(s.v, self.g, s.w, self.gamma)
        s.res.axpby_(1.0, s.u, -1.0, s.v)
        s.xbar.axpby_(1.0, s.x, -float(self.lam), s.res)

    def __iter__(self):
        R = self.x0.dtype.type
        s = DRLSState()
        s.x = self.x0.copy()
        for name in ("u", "v", "w", "res", "res_prev", "xbar", "d", "x_d", "u0", "u1", "temp_x1", "temp_x2"):
            setattr(s, name, s.x.similar())
        s.gamma, s.tau = self.gamma, R(0)
        s.dre = s.res_sq = None
        s.f_u = prox_(s.u, self.f, s.x, self.gamma)
        self._dr_tail(s)
        s.xbar_prev = s.xbar.copy()
        s.H = self.directions.initialize(s.x)
        quasi_newton = hasattr(s.H, "mul_")  # QuasiNewtonStyle: LBFGS, Broyden, AndersonAcceleration
        nesterov = (not quasi_newton) and s.H is not None
        quadratic = is_generalized_quadratic(self.f)
        yield s
        while True:
            dre_curr = s.dre if s.dre is not None else self.DRE(s)
            threshold = R(self.dre_sign * dre_curr - self.c / self.gamma * s.res_sq)  # :163
            if quasi_newton:  # :137-140
                s.H.mul_(s.d, s.res)
                s.d.axpby_(-1.0, s.d)
            elif nesterov:  # :142-144   d = beta (xbar - xbar_prev) + (xbar - x)
                b = R(next(s.H))
                s.d.axpby_(R(R(1) + b), s.xbar, -float(b), s.xbar_prev)
                s.d.axpby_(1.0, s.d, -1.0, s.x)
            else:  # :146-147
                s.d.axpby_(1.0, s.xbar, -1.0, s.x)
            s.x_d.axpby_(1.0, s.x, 1.0, s.d)  # :166
            s.xbar_prev, s.xbar = s.xbar, s.xbar_prev  # :168-169
            s.res_prev, s.res = s.res, s.res_prev
            s.tau = R(1)
            s.x.copy_from(s.x_d)
            s.f_u = prox_(s.u, self.f, s.x, self.gamma)  # :174
            self._dr_tail(s)
            if quasi_newton:  # :151-154
                s.res_prev.axpby_(1.0, s.res, -1.0, s.res_prev)
                s.H.update_(s.d, s.res_prev)
            a = b = c = R(0)
            s.dre = None  # the state moved; the loop below re-evaluates the envelope at every candidate
            for k in range(1, self.max_backtracks + 1):  # :183-195
                if self.dre_sign * self.DRE(s) <= threshold:
                    break
                s.dre = None
                s.tau = R(0) if k == self.max_backtracks else R(s.tau / R(2))
                s.x.axpby_(float(s.tau), s.x_d, float(R(1) - s.tau), s.xbar_prev)
                if quadratic:
                    if k == 1:
                        s.u1.copy_from(s.u)
                        c = prox_(s.u0, self.f, s.xbar_prev, self.gamma)
                        s.temp_x1.axpby_(1.0, s.xbar_prev, -1.0, s.x_d)
                        s.temp_x2.axpby_(1.0, s.xbar_prev, -1.0, s.u0)
                        b = R(s.temp_x1.dot(s.temp_x2) / self.gamma)
                        a = R(s.f_u - b - c)
                    s.u.axpby_(float(s.tau), s.u1, float(R(1) - s.tau), s.u0)
                    s.f_u = R(a * s.tau * s.tau + b * s.tau + c)
                else:
                    s.f_u = prox_(s.u, self.f, s.x, self.gamma)
                self._dr_tail(s)
            yield s


def default_stopping_criterion(tol, iteration, state):
    """norm(state.res, Inf) / state.gamma <= tol  (drls.jl:199-200)"""
    R = state.res.dtype.type
    return state.res.norm_inf() / state.gamma <= R(tol)


def default_solution(iteration, state):
    """drls.jl:201"""
    return state.v


def default_display(it, iteration, state):
    print("%5d | %.3e | %.3e | %.3e" % (it, state.gamma, state.res.norm_inf() / state.gamma, state.tau))


def DRLS(*, maxit=1_000, tol=1e-8, stop=None, solution=default_solution, verbose=False, freq=10, display=default_display,
         **kwargs):
    """drls.jl:235-253"""
    if stop is None:
        stop = lambda iteration, state: default_stopping_criterion(tol, iteration, state)
    return IterativeAlgorithm(DRLSIteration, maxit=maxit, stop=stop, solution=solution, verbose=verbose, freq=freq,
                              display=display, **kwargs)
