"""Li-Lin nonconvex accelerated proximal gradient -- mirror of src/algorithms/li_lin.jl (Algorithm 2 of Li & Lin 2015).

Per iteration: f at z, one value_and_gradient at the extrapolated point y, the prox of g and a handful of AXPBYs; the
monitor branch adds a plain proximal-gradient step from x.  All vector arithmetic runs in the library's HIP kernels.
"""
import numpy as np

from .algorithm import IterativeAlgorithm
from .device import as_hipvector
from .operators import Zero, prox_, value_and_gradient


class LiLinState:
    """li_lin.jl:53-66"""


class LiLinIteration:
    """li_lin.jl:40-49 (f, g, x0, Lf | gamma, adaptive, delta = 1e-3, eta = 0.8); init :69-97; step :99-144.

    li_lin.jl:108 reads an unbound name ``x`` in the monitor branch (UndefVarError in the reference: nothing in that
    branch is executable there).  This mirror follows Algorithm 2 of Li & Lin (2015) in it -- gradient at ``state.x``, and
    for x+ = v the extrapolation y = v + (t/t+)(z - v) + ((t - 1)/t+)(v - x) (li_lin.jl:120-122 has ``z +`` for the paper's
    ``v +``, which diverges on convex problems once the branch fires) -- and counts the visits in ``monitor_branch_taken``."""

    def __init__(self, *, x0, f=None, g=None, Lf=None, gamma=None, adaptive=False, delta=1e-3, eta=0.8, single_sweep=True):
        self.x0 = as_hipvector(x0)
        R = self.x0.dtype.type
        self.f = f if f is not None else Zero()
        self.g = g if g is not None else Zero()
        if gamma is None and Lf is not None:
            gamma = R(1) / R(Lf)
        if gamma is None:
            raise ValueError("either gamma or Lf must be given (the reference would fail on `nothing .* grad`)")
        self.gamma = R(gamma)
        self.adaptive, self.delta, self.eta = bool(adaptive), R(delta), R(eta)
        self.monitor_branch_taken = 0
        self.counters = {"a_passes": 0}
        # f = LeastSquares(A, b) or Composed(loss, A) on a device matrix, g an in-kernel prox kind: one read of A per
        # iteration (two when the monitor branch runs), see _iter_single_sweep
        self._loss_A = _loss_and_matrix(self.f) if single_sweep else None
        if self._loss_A is not None and not (hasattr(self.g, "g_kind") and not (hasattr(self.g, "_scalar") and not self.g._scalar)):
            self._loss_A = None

    def __iter__(self):
        if self._loss_A is not None:
            from . import _lib
            from ._lib import ProxGradError

            gen = self._iter_single_sweep(*self._loss_A)
            try:
                first = next(gen)
            except ProxGradError as e:
                if e.code != _lib.PG_ERR_UNSUPPORTED:  # anything but "shape outside the sweep kernel's range"
                    raise
                self._loss_A = None
                return self._iter_plain()

            def chain():
                yield first
                yield from gen

            return chain()
        return self._iter_plain()

    def _iter_single_sweep(self, loss, A):
        """The same iteration with every product with A folded into the single sweep (pg_mat_fused_tn): the sweep at y gives
        A' grad loss(A y), the prox and A z; f(z) = loss(A z) is an m-vector kernel; A y of the next extrapolated point
        follows by linearity from A z, A x (and A v).  The monitor branch costs one more sweep (at x)."""
        R = self.x0.dtype.type
        s = LiLinState()
        s.y = self.x0.copy()
        s.x = self.x0.copy()
        s.gamma = self.gamma
        n_like, dt, ctx = s.y, s.y.dtype, s.y.ctx
        s.grad_f_y, s.y_forward, s.z, s.res = (n_like.similar() for _ in range(4))
        v, xf, gtmp, rtmp = (n_like.similar() for _ in range(4))
        Ay = A.mul(s.y)
        self.counters["a_passes"] += 1
        Ax = Ay.similar().copy_from(Ay)
        Az, Av, u = Ay.similar(), Ay.similar(), Ay.similar()

        def sweep_at_y():
            s.f_y = R(loss.value_and_gradient(Ay, out=u)[0])
            sc = A.fused_tn(u, s.y, s.gamma, self.g, s.grad_f_y, s.y_forward, s.z, s.res, Az)
            self.counters["a_passes"] += 1
            s.g_z, s.res_inf, s.res_sq = sc[0], sc[1], sc[3]

        sweep_at_y()
        Fy = R(s.f_y + self.g(s.y))
        if not np.isfinite(Fy):
            raise AssertionError("initial point must be feasible")  # :75
        s.theta, s.F_average, s.q = R(1), Fy, R(1)
        yield s
        while True:
            Fz = R(loss(Az) + s.g_z)  # :103
            theta1 = R((R(1) + R(np.sqrt(R(R(1) + R(4) * s.theta * s.theta)))) / R(2))  # :104
            if Fz <= s.F_average - self.delta * s.res_sq:  # :106
                case = 1
            else:
                self.monitor_branch_taken += 1
                loss.value_and_gradient(Ax, out=u)
                sc = A.fused_tn(u, s.x, s.gamma, self.g, gtmp, xf, v, rtmp, Av)  # :108-110 and A v
                self.counters["a_passes"] += 1
                Fv = R(loss(Av) + sc[0])
                case = 1 if Fz <= Fv else 2
            if case == 1:
                c = R((s.theta - R(1)) / theta1)
                s.y.axpby_(R(R(1) + c), s.z, -float(c), s.x)  # :116
                Ay.axpby_(R(R(1) + c), Az, -float(c), Ax)
                s.x, s.z = s.z, s.x
                Ax, Az = Az, Ax
                Fx = Fz
            else:
                c1, c2 = R(s.theta / theta1), R((s.theta - R(1)) / theta1)
                s.y.axpby_(c1, s.z, R(R(1) - c1 + c2), v)  # v + c1 (z - v) + c2 (v - x)   (:120-122, see the class note)
                s.y.axpby_(1.0, s.y, -float(c2), s.x)
                Ay.axpby_(c1, Az, R(R(1) - c1 + c2), Av)
                Ay.axpby_(1.0, Ay, -float(c2), Ax)
                s.x.copy_from(v)
                Ax.copy_from(Av)
                Fx = Fv
            sweep_at_y()  # :128-135
            s.theta = theta1
            q1 = R(self.eta * s.q + R(1))  # :139-141
            s.F_average = R((self.eta * s.q * s.F_average + Fx) / q1)
            s.q = q1
            yield s

    def _iter_plain(self):
        R = self.x0.dtype.type
        s = LiLinState()
        s.y = self.x0.copy()
        s.f_y, g = value_and_gradient(self.f, s.y)
        s.grad_f_y = g.copy() if g is not None else None
        s.gamma = self.gamma
        s.y_forward = s.y.similar().axpby_(1.0, s.y, -float(s.gamma), s.grad_f_y)
        s.z = s.y.similar()
        s.g_z = prox_(s.z, self.g, s.y_forward, s.gamma)
        Fy = R(s.f_y + self.g(s.y))
        if not np.isfinite(Fy):
            raise AssertionError("initial point must be feasible")  # :75
        s.x = self.x0.copy()
        s.res = s.y.similar().axpby_(1.0, s.y, -1.0, s.z)
        s.theta, s.F_average, s.q = R(1), Fy, R(1)
        v, xf = s.y.similar(), s.y.similar()
        yield s
        while True:
            Fz = R(self.f(s.z) + s.g_z)  # :103
            theta1 = R((R(1) + R(np.sqrt(R(R(1) + R(4) * s.theta * s.theta)))) / R(2))  # :104
            if Fz <= s.F_average - self.delta * s.res.norm() ** 2:  # :106
                case = 1
            else:
                self.monitor_branch_taken += 1
                _, gx = value_and_gradient(self.f, s.x)
                xf.axpby_(1.0, s.x, -float(s.gamma), gx)
                g_v = prox_(v, self.g, xf, s.gamma)
                Fv = R(self.f(v) + g_v)
                case = 1 if Fz <= Fv else 2
            if case == 1:
                c = R((s.theta - R(1)) / theta1)
                s.y.axpby_(R(R(1) + c), s.z, -float(c), s.x)  # z + c (z - x)   (:116)
                s.x, s.z = s.z, s.x
                Fx = Fz
            else:
                c1, c2 = R(s.theta / theta1), R((s.theta - R(1)) / theta1)
                s.y.axpby_(c1, s.z, R(R(1) - c1 + c2), v)  # v + c1 (z - v) + c2 (v - x)   (:120-122, see the class note)
                s.y.axpby_(1.0, s.y, -float(c2), s.x)
                s.x.copy_from(v)
                Fx = Fv
            s.f_y, g = value_and_gradient(self.f, s.y)  # :128
            s.grad_f_y.copy_from(g)
            s.y_forward.axpby_(1.0, s.y, -float(s.gamma), s.grad_f_y)
            s.g_z = prox_(s.z, self.g, s.y_forward, s.gamma)
            s.res.axpby_(1.0, s.y, -1.0, s.z)
            s.theta = theta1
            q1 = R(self.eta * s.q + R(1))  # :139-141
            s.F_average = R((self.eta * s.q * s.F_average + Fx) / q1)
            s.q = q1
            yield s


from ._composed import loss_and_matrix as _loss_and_matrix  # noqa: E402


def default_stopping_criterion(tol, iteration, state):
    """norm(state.res, Inf) / state.gamma <= tol  (li_lin.jl:146-147)"""
    R = state.res.dtype.type
    res_inf = state.res_inf if getattr(state, "res_inf", None) is not None else state.res.norm_inf()
    return R(res_inf) / state.gamma <= R(tol)


def default_solution(iteration, state):
    """li_lin.jl:148"""
    return state.z


def default_display(it, iteration, state):
    print("%5d | %.3e | %.3e" % (it, state.gamma, state.res.norm_inf() / state.gamma))


def LiLin(*, maxit=10_000, tol=1e-8, stop=None, solution=default_solution, verbose=False, freq=100,
          display=default_display, **kwargs):
    """li_lin.jl:183-201"""
    if stop is None:
        stop = lambda iteration, state: default_stopping_criterion(tol, iteration, state)
    return IterativeAlgorithm(LiLinIteration, maxit=maxit, stop=stop, solution=solution, verbose=verbose, freq=freq,
                              display=display, **kwargs)
