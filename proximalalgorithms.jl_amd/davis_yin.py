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

    def __init__(self, *, x0, f=None, g=None, h=None, lam=1.0, Lf=None, gamma=None, single_sweep=True, **kw):
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
        self.counters = {"a_passes": 0}
        # f = LeastSquares / Composed(loss, A) on a device matrix, g and h prox kinds the sweep applies per column: ONE read
        # of A per iteration (_iter_single_sweep)
        from ._composed import loss_and_matrix

        la = loss_and_matrix(self.f) if single_sweep else None
        specs = (self._prox_spec(self.g), self._prox_spec(self.h))
        self._sweep = (la[0], la[1], specs[0], specs[1]) if (la is not None and None not in specs) else None

    @property
    def graph_safe(self):
        """constant gamma / lambda and no buffer swaps in the plain body; the single-sweep form swaps and is not recorded"""
        return self._sweep is None

    @staticmethod
    def _prox_spec(op):
        """(kind, p0, p1) of an operator the sweep kernel can apply per column, else None"""
        from ._lib import PG_G_SQRNORML2
        from .operators import IndBox, NormL1, SqrNormL2

        if isinstance(op, SqrNormL2):
            return PG_G_SQRNORML2, op.lam, 0.0
        if isinstance(op, (IndBox, NormL1)) and not op._scalar:
            return None
        if isinstance(op, (NormL1, IndBox, Zero)):
            p0, p1 = op.g_params()
            return op.g_kind, p0, p1
        return None

    def _iter_single_sweep(self, loss, A, g_spec, h_spec):
        """davis_yin.jl:62-84 for f = loss o A with every product folded into ONE read of A per iteration
        (pg_mat_fused_dys): the sweep takes r = grad loss(A xg) and returns the gradient, z_half, xh, res, the updated z
        and already the next prox_g point with its image A xg -- whose loss gradient is the next sweep's r."""
        s = DavisYinState(self.x0.copy())
        xg_next, z_next = s.z.similar(), s.z.similar()
        prox_(s.xg, self.g, s.z, self.gamma, want_value=False)  # :64 / :74 of the first iteration
        Axg = A.mul(s.xg)
        self.counters["a_passes"] += 1
        Axg_next, u = Axg.similar(), Axg.similar()
        while True:
            loss.value_and_gradient(Axg, out=u)
            sc = A.fused_dys(u, s.xg, s.z, self.gamma, self.lam, g_spec, h_spec, s.grad_f_xg, s.z_half, s.xh, s.res, z_next,
                             xg_next, Axg_next)
            self.counters["a_passes"] += 1
            s.z, z_next = z_next, s.z  # :80
            s.res_inf = sc[0]
            yield s
            # the next iteration's prox!(xg, g, z) (:74) has already been applied by the sweep
            s.xg, xg_next = xg_next, s.xg
            Axg, Axg_next = Axg_next, Axg

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
        if self._sweep is not None:
            from . import _lib
            from ._lib import ProxGradError

            gen = self._iter_single_sweep(*self._sweep)
            try:
                first = next(gen)
            except ProxGradError as e:
                if e.code != _lib.PG_ERR_UNSUPPORTED:  # anything but "shape outside the sweep kernel's range"
                    raise
                self._sweep = None
            else:
                def chain():
                    yield first
                    yield from gen

                return chain()
        return self._iter_plain()

    def _iter_plain(self):
        s = self.init_state()
        while True:
            self.body(s)
            yield s


def default_stopping_criterion(tol, iteration, state):
    """norm(state.res, Inf) <= tol  (davis_yin.jl:86-87)"""
    res_inf = state.res_inf if getattr(state, "res_inf", None) is not None else state.res.norm_inf()
    return res_inf <= state.res.dtype.type(tol)


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
