"""Davis-Yin three-operator splitting -- mirror of src/algorithms/davis_yin.jl.

minimize f(x) + g(x) + h(x), f smooth: per iteration prox_g, one gradient of f, prox_h and two AXPBYs, all on the device.
"""
from .algorithm import IterativeAlgorithm
from .device import as_hipvector
from .operators import Zero, prox_, value_and_gradient_


class DavisYinState:
    """davis_yin.jl:53-60"""

    def __init__(self, z):
        self.z = z
        self.xg, self.grad_f_xg, self.z_half, self.xh, self.res = (z.similar() for _ in range(5))


class DavisYinIteration:
    """davis_yin.jl:41-50 (f, g, h, x0, lambda = 1, Lf | gamma = 1 / Lf) and Base.iterate :62-84"""

    def __init__(self, *, x0, f=None, g=None, h=None, lam=1.0, Lf=None, gamma=None, **kw):
        if "lambda_" in kw:
            lam = kw.pop("lambda_")
        if kw:
            raise TypeError(f"unexpected keyword arguments {sorted(kw)}")
        self.x0 = as_hipvector(x0)
        R = self.x0.dtype.type
        self.f, self.g, self.h = (o if o is not None else Zero() for o in (f, g, h))
        if gamma is None:
            if Lf is None:
                raise ValueError("You must specify either Lf or gamma")  # :48-49
            gamma = R(1) / R(Lf)
        self.gamma, self.lam = R(gamma), R(lam)

    graph_safe = True  # constant gamma / lambda, no buffer swaps

    def init_state(self):
        return DavisYinState(self.x0.copy())

    def body(self, s):
        """one Base.iterate (davis_yin.jl:73-83), allocation-free"""
        gamma = self.gamma
        prox_(s.xg, self.g, s.z, gamma, want_value=False)  # :74
        value_and_gradient_(s.grad_f_xg, self.f, s.xg)  # :75-76
        s.z_half.axpby_(2.0, s.xg, -1.0, s.z)  # :77  2 xg - z - gamma grad
        s.z_half.axpby_(1.0, s.z_half, -float(gamma), s.grad_f_xg)
        prox_(s.xh, self.h, s.z_half, gamma, want_value=False)  # :78
        s.res.axpby_(1.0, s.xh, -1.0, s.xg)  # :79
        s.z.axpby_(1.0, s.z, float(self.lam), s.res)  # :80

    def __iter__(self):
        s = self.init_state()
        while True:
            self.body(s)
            yield s


def default_stopping_criterion(tol, iteration, state):
    """norm(state.res, Inf) <= tol  (davis_yin.jl:86-87)"""
    return state.res.norm_inf() <= state.res.dtype.type(tol)


def default_solution(iteration, state):
    """davis_yin.jl:88"""
    return state.xh


def default_display(it, iteration, state):
    print("%5d | %.3e" % (it, state.res.norm_inf()))


def DavisYin(*, maxit=10_000, tol=1e-8, stop=None, solution=default_solution, verbose=False, freq=100,
             display=default_display, graph=False, **kwargs):
    """davis_yin.jl:114-132.  graph=True: the iteration body is recorded into a hipGraph after two plain iterations and
    replayed with one launch per iteration."""
    if stop is None:
        stop = lambda iteration, state: default_stopping_criterion(tol, iteration, state)
    return IterativeAlgorithm(DavisYinIteration, maxit=maxit, stop=stop, solution=solution, verbose=verbose, freq=freq,
                              display=display, graph=graph, **kwargs)
