"""Primal-dual splittings -- mirror of src/algorithms/primal_dual.jl: AFBA (asymmetric forward-backward-adjoint,
Latafat & Patrinos 2017), Vu-Condat (theta = 2) and Chambolle-Pock (theta = 2, f = Zero, l = IndZero).

    minimize f(x) + g(x) + (h [] l)(L x)

Per iteration: gradient of f, prox of g, gradient of l*, prox of h* (Moreau), and four products with L / L' -- GEMV passes
over the device matrix; every vector statement is a HIP kernel of the library.
"""
import math

import numpy as np

from .algorithm import IterativeAlgorithm
from .device import HIPMatrix, as_hipvector
from .operators import IndZero, Zero, convex_conjugate, prox_, value_and_gradient_


def _isapprox(a, b, R):
    return abs(float(a) - float(b)) <= math.sqrt(np.finfo(R).eps) * max(abs(float(a)), abs(float(b)))


def AFBA_default_stepsizes(nmL, h, theta, mu, beta_f, beta_l, R=np.float64):
    """primal_dual.jl:334-416 with ``nmL`` = opnorm(L).  A balance ``alpha`` between primal and dual steps is chosen
    from the relative sizes of ||L||, beta_f and beta_l; then gamma1 = 1 / (beta_f / 2 + c1 ||L|| / alpha) and
    gamma2 = 0.99 / (beta_l / 2 + c2 ||L|| alpha) with method-dependent factors."""
    R = np.dtype(R).type
    theta, mu, beta_f, beta_l, nmL = R(theta), R(mu), R(beta_f), R(beta_l), R(nmL)
    if isinstance(h, Zero):
        return R(1.99) / beta_f, R(1)
    par, par2 = R(5), R(100)

    def balance(n):
        if n > par * max(beta_l, beta_f):
            return R(1)
        if beta_f > par * beta_l:
            return par2 * n / beta_f
        if beta_l > par * beta_f:
            return beta_l / (par2 * n)
        return R(1)

    def steps(n, alpha, c1=R(1), c2=R(1)):
        return R(R(1) / (beta_f / 2 + c1 * n / alpha)), R(R(0.99) / (beta_l / 2 + c2 * n * alpha))

    if _isapprox(theta, 2, R):  # Vu-Condat
        return steps(nmL, balance(nmL))
    if _isapprox(theta, 1, R) and _isapprox(mu, 1, R):  # SPCA
        alpha = R(1)
        if not nmL > par2 * beta_l and beta_l > par * beta_f:
            alpha = beta_l / (par2 * nmL)
        g1 = R(1.99) / beta_f if beta_f > 0 else R(1) / (nmL / alpha)
        return R(g1), R(R(0.99) / (beta_l / 2 + g1 * nmL * nmL))
    if _isapprox(theta, 0, R) and _isapprox(mu, 1, R):  # PPCA
        if _isapprox(beta_f, 0, R):
            n = R(nmL * R(np.sqrt(R(3))))
            return steps(n, R(1) if n > par * beta_l else beta_l / (par2 * n))
        alpha = balance(nmL)
        return steps(nmL, alpha, c2=1 + 2 * nmL / (nmL + alpha * beta_f / 2))
    if _isapprox(mu, 0, R):  # SDCA, PDCA
        temp = theta * theta - 3 * theta + 3
        if _isapprox(beta_l, 0, R):
            n = R(nmL * R(np.sqrt(temp)))
            return steps(n, R(1) if n > par * beta_f else par2 * n / beta_f)
        alpha = balance(nmL)
        return steps(nmL, alpha, c1=1 + (temp - 1) * alpha * nmL / (alpha * nmL + beta_l / 2))
    if _isapprox(theta, 0, R) and _isapprox(mu, 0.5, R):  # PPDCA
        if _isapprox(beta_l, 0, R) or _isapprox(beta_f, 0, R):
            alpha = balance(nmL)
        else:
            alpha = R(np.sqrt(beta_l / beta_f)) / 2
        return steps(nmL, alpha)
    raise ValueError("this choice of theta and mu is not supported!")


def _opnorm(L, iters=200):
    """opnorm(L): exact (host SVD) for small operators, power iteration on L'L with the device GEMVs otherwise"""
    if L.m * L.n <= (1 << 22):
        return float(np.linalg.norm(L.numpy().astype(np.float64), 2))
    from .device import HIPVector

    v = HIPVector.empty(L.n, L.dtype, L.ctx).fill_(1.0 / math.sqrt(L.n))
    u = HIPVector.empty(L.m, L.dtype, L.ctx)
    nrm = 1.0
    for _ in range(iters):
        L.mul(v, u)
        L.mul_adjoint(u, v)
        nrm = float(v.norm())
        v.axpby_(1.0 / nrm, v)
    return math.sqrt(nrm)


class AFBAState:
    """primal_dual.jl:158-169"""

    def __init__(self, x, y):
        self.x, self.y = x, y
        self.xbar, self.gradf, self.FPR_x, self.temp_x = (x.similar() for _ in range(4))
        self.ybar, self.gradl, self.FPR_y, self.temp_y = (y.similar() for _ in range(4))


class AFBAIteration:
    """primal_dual.jl:83-112 (options) and Base.iterate :171-209.  ``L``: HIPMatrix / numpy matrix, or None for the
    identity (0 * I when h is Zero, :87-91)."""

    def __init__(self, *, x0, y0, f=None, g=None, h=None, l=None, L=None, beta_f=None, beta_l=None, theta=1.0, mu=1.0,
                 lam=1.0, gamma=None, opnorm_L=None, single_sweep=True, **kw):
        if "lambda_" in kw:
            lam = kw.pop("lambda_")
        if kw:
            raise TypeError(f"unexpected keyword arguments {sorted(kw)}")
        self.x0 = as_hipvector(x0)
        self.y0 = as_hipvector(y0, self.x0.ctx)
        R = self.x0.dtype.type
        self.f, self.g, self.h = (o if o is not None else Zero() for o in (f, g, h))
        self.l = l if l is not None else IndZero()
        if L is not None and not isinstance(L, HIPMatrix):
            L = HIPMatrix.from_numpy(np.asfortranarray(np.asarray(L, dtype=self.x0.dtype)), self.x0.ctx)
        self.L = L
        self._zero_L = L is None and isinstance(self.h, Zero)
        if beta_f is None:
            if not isinstance(self.f, Zero):
                raise ValueError("argument beta_f must be specified together with f")  # :96
            beta_f = 0
        if beta_l is None:
            if not isinstance(self.l, IndZero):
                raise ValueError("argument beta_l must be specified together with l")  # :101
            beta_l = 0
        self.theta, self.mu, self.lam = R(theta), R(mu), R(lam)
        if gamma is None:
            if self.lam != 1:
                raise ValueError("if lambda != 1, then you need to provide stepsizes manually")  # :107
            if opnorm_L is None:
                opnorm_L = 0.0 if self._zero_L else (1.0 if L is None else _opnorm(L))
            gamma = AFBA_default_stepsizes(opnorm_L, self.h, theta, mu, beta_f, beta_l, R)
        self.gamma = (R(gamma[0]), R(gamma[1]))
        # Vu-Condat / Chambolle-Pock with a device matrix L and an in-kernel prox kind for g: ONE read of L per iteration
        # (see _body_single_sweep)
        self.single_sweep = bool(single_sweep) and isinstance(self.L, HIPMatrix) \
            and ((self.theta == 2 and self.lam == 1) or (self.theta == 1 and self.mu == 1)) \
            and hasattr(self.g, "g_kind") and not (hasattr(self.g, "_scalar") and not self.g._scalar)
        self.counters = {"L_passes": 0}

    def _mul(self, out, x):
        if self._zero_L:
            return out.fill_(0.0)
        if self.L is None:
            return out.copy_from(x)
        self.counters["L_passes"] += 1
        return self.L.mul(x, out)

    def _mul_adjoint(self, out, y):
        if self._zero_L:
            return out.fill_(0.0)
        if self.L is None:
            return out.copy_from(y)
        self.counters["L_passes"] += 1
        return self.L.mul_adjoint(y, out)

    graph_safe = True  # constant step sizes, no buffer swaps: the body can be recorded once and replayed (hipGraph)

    def init_state(self):
        s = AFBAState(self.x0.copy(), self.y0.copy())
        s.hc, s.lc = convex_conjugate(self.h), convex_conjugate(self.l)
        # gradients of a Zero term stay zero: written once here instead of once per iteration
        s.f_zero, s.lc_zero = isinstance(self.f, Zero), isinstance(s.lc, Zero)
        s.gradf.fill_(0.0)
        s.gradl.fill_(0.0)
        return s

    def _body_single_sweep(self, s):
        """theta = 2, lambda = 1 (Vu-Condat / Chambolle-Pock), L a device matrix: one read of L per iteration (theta = 1,
        mu = 1 -- the default AFBA -- two: the sweep and the primal correction L'(gamma1 FPR_y)).
        The sweep pg_mat_fused_tn takes the dual iterate y as its m-vector and returns  L'y,  xbar = prox_{g1 g}(x - g1
        (L'y + grad f))  (the smooth term enters by shifting the sweep's x) and  L xbar;  with lambda = 1 the next primal
        iterate IS xbar, so  L (2 xbar - x) = 2 L xbar - L x  needs no product: L x is the previous sweep's L xbar.  The
        two correction products of primal_dual.jl:199-205 carry the factor (2 - theta) = 0.  Returns False (nothing
        written) when the sweep kernel does not cover this matrix."""
        from . import _lib
        from ._lib import ProxGradError

        g1, g2 = self.gamma
        if getattr(s, "Lx", None) is None:  # first use: workspaces and L x0
            s.Lx, s.Lxbar = s.y.similar(), s.y.similar()
            s.sw_y, s.sw_res, s.sw_x = s.x.similar(), s.x.similar(), s.x.similar()
            s.Lx_valid = False
        if not s.f_zero:
            value_and_gradient_(s.gradf, self.f, s.x)  # :180
            xs = s.sw_x.axpby_(1.0, s.x, -float(g1), s.gradf)
        else:
            xs = s.x
        try:
            self.L.fused_tn(s.y, xs, g1, self.g, s.temp_x, s.sw_y, s.xbar, s.sw_res, s.Lxbar)  # :182-186 (+ L xbar)
        except ProxGradError as e:
            if e.code == _lib.PG_ERR_UNSUPPORTED:
                return False
            raise
        self.counters["L_passes"] += 1
        if self.theta == 2 and not s.Lx_valid:
            self._mul(s.Lx, s.x)  # once: L x0
            s.Lx_valid = True
        if not s.lc_zero:
            value_and_gradient_(s.gradl, s.lc, s.y)  # :187
        if self.theta == 2:
            s.temp_y.axpby_(2.0, s.Lxbar, -1.0, s.Lx)  # :189-190  L (2 xbar - x)
            s.temp_y.axpby_(1.0, s.temp_y, -1.0, s.gradl)  # :191
        else:  # theta = 1: L (theta xbar + (1 - theta) x) = L xbar, straight out of the sweep
            s.temp_y.axpby_(1.0, s.Lxbar, -1.0, s.gradl)
        s.temp_y.axpby_(float(g2), s.temp_y, 1.0, s.y)  # :192-193
        prox_(s.ybar, s.hc, s.temp_y, g2, want_value=False)  # :194
        s.FPR_x.axpby_(1.0, s.xbar, -1.0, s.x)  # :196-197
        s.FPR_y.axpby_(1.0, s.ybar, -1.0, s.y)
        if self.theta == 2:
            s.x.axpby_(1.0, s.x, 1.0, s.FPR_x)  # :201 with a zero correction, lambda = 1
            s.Lx.copy_from(s.Lxbar)  # L x of the new x (= xbar up to the rounding of x + (xbar - x))
            s.y.axpby_(1.0, s.y, 1.0, s.FPR_y)  # :205
        else:  # theta = 1, mu = 1: the primal correction L'(gamma1 FPR_y) is a real product (:199-201), the dual one vanishes
            s.temp_y.axpby_(float(g1), s.FPR_y)
            self._mul_adjoint(s.temp_x, s.temp_y)
            s.temp_x.axpby_(1.0, s.FPR_x, -1.0, s.temp_x)
            s.x.axpby_(1.0, s.x, float(self.lam), s.temp_x)
            s.y.axpby_(1.0, s.y, float(self.lam), s.FPR_y)  # :205
        return True

    def body(self, s):
        """one Base.iterate (primal_dual.jl:176-209), allocation-free"""
        if self.single_sweep:
            if self._body_single_sweep(s):
                return
            self.single_sweep = False  # outside the sweep kernel's range: the plain statement order from now on
        R = self.x0.dtype.type
        g1, g2 = self.gamma
        theta, mu, lam = self.theta, self.mu, self.lam
        if not s.f_zero:
            value_and_gradient_(s.gradf, self.f, s.x)  # :180
        self._mul_adjoint(s.temp_x, s.y)  # :182-185   x - gamma1 (L'y + grad f)
        s.temp_x.axpby_(1.0, s.temp_x, 1.0, s.gradf)
        s.temp_x.axpby_(-float(g1), s.temp_x, 1.0, s.x)
        prox_(s.xbar, self.g, s.temp_x, g1, want_value=False)  # :186
        if not s.lc_zero:
            value_and_gradient_(s.gradl, s.lc, s.y)  # :187
        s.temp_x.axpby_(float(theta), s.xbar, float(R(1) - theta), s.x)  # :189
        self._mul(s.temp_y, s.temp_x)  # :190-193   y + gamma2 (L t - grad l*)
        s.temp_y.axpby_(1.0, s.temp_y, -1.0, s.gradl)
        s.temp_y.axpby_(float(g2), s.temp_y, 1.0, s.y)
        prox_(s.ybar, s.hc, s.temp_y, g2, want_value=False)  # :194
        s.FPR_x.axpby_(1.0, s.xbar, -1.0, s.x)  # :196-197
        s.FPR_y.axpby_(1.0, s.ybar, -1.0, s.y)
        # :199-205.  The correction terms L'(c1 FPR_y) and L(c2 FPR_x) carry the factors mu (2 - theta) gamma1 and
        # (1 - mu)(2 - theta) gamma2: exactly zero for Vu-Condat / Chambolle-Pock (theta = 2) and, one each, for mu = 0 or 1 --
        # the product with L is skipped then (the reference multiplies a zero vector): 2 or 3 reads of L instead of 4
        c1 = R(mu * (R(2) - theta) * g1)
        if c1 != 0:
            s.temp_y.axpby_(float(c1), s.FPR_y)
            self._mul_adjoint(s.temp_x, s.temp_y)
            s.temp_x.axpby_(1.0, s.FPR_x, -1.0, s.temp_x)
            s.x.axpby_(1.0, s.x, float(lam), s.temp_x)
        else:
            s.x.axpby_(1.0, s.x, float(lam), s.FPR_x)
        c2 = R((R(1) - mu) * (R(2) - theta) * g2)
        if c2 != 0:
            s.temp_x.axpby_(float(c2), s.FPR_x)
            self._mul(s.temp_y, s.temp_x)
            s.temp_y.axpby_(1.0, s.FPR_y, 1.0, s.temp_y)
            s.y.axpby_(1.0, s.y, float(lam), s.temp_y)
        else:
            s.y.axpby_(1.0, s.y, float(lam), s.FPR_y)

    def __iter__(self):
        s = self.init_state()
        while True:
            self.body(s)
            yield s


def VuCondatIteration(**kwargs):
    """primal_dual.jl:131"""
    kwargs["theta"] = 2
    return AFBAIteration(**kwargs)


def ChambollePockIteration(**kwargs):
    """primal_dual.jl:151-152"""
    kwargs.update(theta=2, f=Zero(), l=IndZero())
    return AFBAIteration(**kwargs)


def default_stopping_criterion(tol, iteration, state):
    """norm(FPR_x, Inf) + norm(FPR_y, Inf) <= tol  (primal_dual.jl:211-212)"""
    R = state.x.dtype.type
    return R(state.FPR_x.norm_inf() + state.FPR_y.norm_inf()) <= R(tol)


def default_solution(iteration, state):
    """primal_dual.jl:213"""
    return state.xbar, state.ybar


def default_display(it, iteration, state):
    print("%6d | %.3e" % (it, state.FPR_x.norm_inf() + state.FPR_y.norm_inf()))


def _make(iterator_type, maxit, tol, stop, solution, verbose, freq, display, graph, kwargs):
    if stop is None:
        stop = lambda iteration, state: default_stopping_criterion(tol, iteration, state)
    return IterativeAlgorithm(iterator_type, maxit=maxit, stop=stop, solution=solution, verbose=verbose, freq=freq,
                              display=display, graph=graph, **kwargs)


def AFBA(*, maxit=10_000, tol=1e-5, stop=None, solution=default_solution, verbose=False, freq=100, graph=False,
         display=default_display, **kwargs):
    """primal_dual.jl:250-268.  graph=True: after two plain iterations the body is recorded into a hipGraph and every
    further iteration is ONE graph launch plus the stop test (same arithmetic, same iterates)."""
    return _make(AFBAIteration, maxit, tol, stop, solution, verbose, freq, display, graph, kwargs)


def VuCondat(*, maxit=10_000, tol=1e-5, stop=None, solution=default_solution, verbose=False, freq=100, graph=False,
             display=default_display, **kwargs):
    """primal_dual.jl:297-298: AFBA with theta = 2"""
    return _make(VuCondatIteration, maxit, tol, stop, solution, verbose, freq, display, graph, kwargs)


def ChambollePock(*, maxit=10_000, tol=1e-5, stop=None, solution=default_solution, verbose=False, freq=100, graph=False,
                  display=default_display, **kwargs):
    """primal_dual.jl:328-329: AFBA with theta = 2, f = Zero, l = IndZero"""
    return _make(ChambollePockIteration, maxit, tol, stop, solution, verbose, freq, display, graph, kwargs)
