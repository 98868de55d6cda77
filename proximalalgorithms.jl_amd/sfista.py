"""SFISTA -- mirror of src/algorithms/sfista.jl (strongly convex FISTA variant, Kong 2021 Alg. 2.2.2).

Per iteration: one gradient evaluation at the prox centre ``xt`` (two GEMV sweeps for LeastSquares), the prox of g, and
three AXPBY-type updates; the default stopping rule (sfista.jl:101-107) needs one more gradient at ``y``.  All vector
arithmetic runs in the library's HIP kernels.
"""
import math

import numpy as np

from .algorithm import IterativeAlgorithm
from .device import as_hipvector
from .operators import Zero, prox_, value_and_gradient


class SFISTAState:
    """sfista.jl:51-63"""

    def __init__(self, lam, yPrev):
        R = yPrev.dtype.type
        self.lam = lam
        self.yPrev = yPrev
        self.y = yPrev.similar().fill_(0.0)
        self.xPrev = yPrev.copy()
        self.x = yPrev.similar().fill_(0.0)
        self.xt = yPrev.similar().fill_(0.0)
        self.tau, self.a, self.APrev, self.A = R(1), R(0), R(1), R(0)
        self.gradf_xt = yPrev.similar().fill_(0.0)
        self.tmp = yPrev.similar()


class SFISTAIteration:
    """sfista.jl:37-47 (x0, f, g, Lf, mf = 0) and Base.iterate :65-92"""

    REFRESH = 64  # single sweep: A x is carried by its linear recurrence and recomputed exactly every REFRESH iterations

    def __init__(self, *, x0, f=None, g=None, Lf, mf=0.0, single_sweep=True):
        self.x0 = as_hipvector(x0)
        R = self.x0.dtype.type
        self.f = f if f is not None else Zero()
        self.g = g if g is not None else Zero()
        self.Lf, self.mf = R(Lf), R(mf)
        self.counters = {"a_passes": 0}
        # f = LeastSquares(A, b) / Composed(loss, A) on a device matrix, g an in-kernel prox kind: the gradient at xt, the
        # prox and A y in ONE read of A (+ one for the default stop rule's gradient at y): 2 reads per iteration, not 4
        from .li_lin import _loss_and_matrix

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
        """sfista.jl:65-92 with the products folded into the sweep pg_mat_fused_tn: at xt it returns A' grad loss(A xt), the
        prox point y and A y; A xt and A x follow from the (linear) recurrences of xt and x -- A y enters fresh from every
        sweep, A x is recomputed exactly every REFRESH iterations so rounding cannot accumulate."""
        R = self.x0.dtype.type
        s = SFISTAState(R(1) / self.Lf, self.x0.copy())
        mf = self.mf
        AyPrev = A.mul(s.yPrev)
        self.counters["a_passes"] += 1
        AxPrev = AyPrev.similar().copy_from(AyPrev)
        Ax, Axt, u = AyPrev.similar(), AyPrev.similar(), AyPrev.similar()
        s.Ay, s.u_tmp, s.loss_A = AyPrev.similar(), AyPrev.similar(), (loss, A)
        fwd, rtmp, s.g_tmp = s.x.similar(), s.x.similar(), s.x.similar()
        k = 0
        while True:
            s.tau = R(s.lam * (R(1) + mf * s.APrev))  # :70
            s.a = R((s.tau + R(np.sqrt(R(s.tau * s.tau + R(4) * s.tau * s.APrev)))) / R(2))  # :71
            s.A = R(s.APrev + s.a)  # :72
            cy, cx = R(s.APrev / s.A), R(s.a / s.A)
            s.xt.axpby_(cy, s.yPrev, cx, s.xPrev)  # :73
            Axt.axpby_(cy, AyPrev, cx, AxPrev)
            loss.value_and_gradient(Axt, out=u)  # :74 (grad loss at A xt)
            lam2 = R(s.lam / (R(1) + s.lam * mf))  # :76
            A.fused_tn(u, s.xt, lam2, self.g, s.gradf_xt, fwd, s.y, rtmp, s.Ay)  # :74-78 and A y
            self.counters["a_passes"] += 1
            c = R(s.a / (R(1) + s.A * mf))
            c0, c1, c2 = R(R(1) - c * mf), R(c * (R(1) / s.lam + mf)), R(c / s.lam)
            s.x.axpby_(c0, s.xPrev, c1, s.y)  # :79-82
            s.x.axpby_(1.0, s.x, -float(c2), s.xt)
            k += 1
            if k % self.REFRESH == 0:
                A.mul(s.x, Ax)
                self.counters["a_passes"] += 1
            else:
                Ax.axpby_(c0, AxPrev, c1, s.Ay)
                Ax.axpby_(1.0, Ax, -float(c2), Axt)
            s.yPrev.copy_from(s.y)  # :84-86
            s.xPrev.copy_from(s.x)
            AyPrev.copy_from(s.Ay)
            AxPrev.copy_from(Ax)
            s.APrev = s.A
            yield s

    def _iter_plain(self):
        R = self.x0.dtype.type
        s = SFISTAState(R(1) / self.Lf, self.x0.copy())
        mf = self.mf
        while True:
            s.tau = R(s.lam * (R(1) + mf * s.APrev))  # :70
            s.a = R((s.tau + R(np.sqrt(R(s.tau * s.tau + R(4) * s.tau * s.APrev)))) / R(2))  # :71
            s.A = R(s.APrev + s.a)  # :72
            s.xt.axpby_(R(s.APrev / s.A), s.yPrev, R(s.a / s.A), s.xPrev)  # :73
            _, g = value_and_gradient(self.f, s.xt)  # :74
            s.gradf_xt.copy_from(g)
            lam2 = R(s.lam / (R(1) + s.lam * mf))  # :76
            s.tmp.axpby_(1.0, s.xt, -float(lam2), s.gradf_xt)
            prox_(s.y, self.g, s.tmp, lam2, want_value=False)  # :78
            # x = xPrev + c ((y - xt) / lam + mf (y - xPrev))   (:79-82), regrouped per vector
            c = R(s.a / (R(1) + s.A * mf))
            s.x.axpby_(R(R(1) - c * mf), s.xPrev, R(c * (R(1) / s.lam + mf)), s.y)
            s.x.axpby_(1.0, s.x, -float(R(c / s.lam)), s.xt)
            s.yPrev.copy_from(s.y)  # :84-86
            s.xPrev.copy_from(s.x)
            s.APrev = s.A
            yield s


def check_sc(state, iteration, tol, termination_type=""):
    """sfista.jl:95-108: returns (res, stop).  Only the classic criterion is defined by the reference's fields (the
    "AIPP" branch reads ``iter.y0``, which SFISTAIteration does not have)."""
    if termination_type == "AIPP":
        raise AttributeError("SFISTAIteration has no field y0 (sfista.jl:98-100)")
    R = state.y.dtype.type
    lam2 = R(state.lam / (R(1) + state.lam * iteration.mf))
    if getattr(state, "Ay", None) is not None:  # single sweep: A y is known, the gradient at y costs one read of A
        loss, A = state.loss_A
        loss.value_and_gradient(state.Ay, out=state.u_tmp)
        gy = A.mul_adjoint(state.u_tmp, state.g_tmp)
        iteration.counters["a_passes"] += 1
    else:
        _, gy = value_and_gradient(iteration.f, state.y)
    r = state.tmp
    r.axpby_(1.0, gy, -1.0, state.gradf_xt)  # grad f(y) - grad f(xt)
    r.axpby_(1.0, r, float(R(1) / lam2), state.xt)
    r.axpby_(1.0, r, -float(R(1) / lam2), state.y)
    res = r.norm()
    tol = R(tol)
    approx = abs(float(res) - float(tol)) <= math.sqrt(np.finfo(R).eps) * max(abs(float(res)), abs(float(tol)))
    return res, bool(res <= tol or approx)


def default_solution(iteration, state):
    """sfista.jl:111"""
    return state.y


def default_display(it, iteration, state):
    print("%5d | %.3e" % (it, check_sc(state, iteration, 0.0)[0]))


def SFISTA(*, maxit=10_000, tol=1e-6, termination_type="", stop=None, solution=default_solution, verbose=False, freq=100,
           display=default_display, **kwargs):
    """sfista.jl:146-166"""
    if stop is None:
        stop = lambda iteration, state: check_sc(state, iteration, tol, termination_type)[1]
    return IterativeAlgorithm(SFISTAIteration, maxit=maxit, stop=stop, solution=solution, verbose=verbose, freq=freq,
                              display=display, **kwargs)
