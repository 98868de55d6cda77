"""Single-sweep ForwardBackward / FastForwardBackward for compositions x -> loss(A x) (engine "composed").

The reference evaluates the smooth term of e.g. sparse logistic regression (test_sparse_logistic_small.jl:20-30,
BASELINE config 4's objective) through value_and_gradient(f, x) = (loss(A x), A' grad loss(A x)): two reads of A per
gradient, one more per line-search trial (f(z)).  For f = Composed(loss, A) with a device matrix the iteration body is
re-associated so that A is read ONCE per iteration, exactly like the LeastSquares path:

  * one sweep (pg_mat_fused_tn) forms  grad_f_x = A' u  (u = grad loss(A x), an m-vector),  y = x - gamma grad_f_x,
    z = prox_{gamma g}(y),  res = x - z  AND  A z  while each column is in registers;
  * the next point's image needs no pass:  FB: x+ = z, so A x+ = A z;  FFB: x+ = z + beta (z - z_prev), so
    A x+ = (1 + beta) A z - beta A z_prev  (each A z comes fresh out of a sweep -- nothing accumulates);
  * the line search's f(z) = loss(A z) is an m-vector kernel on the A z the sweep left behind; only a REJECTED trial
    costs a pass (A z for the shrunk step).

Same statements, same order of decisions as forward_backward.jl:65-123 / fast_forward_backward.jl:73-145 and
fb_tools.jl:24-63; the iterates equal the generic engine's up to the rounding of the re-association.
"""
import warnings

import numpy as np

from . import _lib
from ._lib import ProxGradError
from .device import HIPMatrix, HIPVector
from .fb_tools import f_model, lower_bound_smoothness_constant
from .nesterov import AdaptiveNesterovSequence
from .operators import Composed, prox_


def loss_and_matrix(f):
    """(loss on m-vectors, device matrix) when f(x) = loss(A x) with A a HIPMatrix -- Composed(loss, A), or an unsharded
    LeastSquares(A, b, lam) read as lam/2 ||. - b||^2 after A -- else None"""
    from .operators import LeastSquares, SquaredDistance

    if isinstance(f, Composed) and isinstance(f.A, HIPMatrix):
        return f.f, f.A
    if isinstance(f, LeastSquares) and f.comm is None:
        return SquaredDistance(f.b, lam=f.lam), f.A
    return None


def composed_supported(f, g):
    """f(x) = loss(A x) on a device matrix, g one of the prox kinds the sweep applies in-kernel"""
    if loss_and_matrix(f) is None or not hasattr(g, "g_kind"):
        return False
    return not (hasattr(g, "_scalar") and not g._scalar)


class _Sweep:
    """Vectors and bookkeeping shared by the two iterations."""

    def __init__(self, it, state_cls, with_prev):
        self.it = it
        f, R = it.f, it.x0.dtype.type
        self.loss, self.A = loss_and_matrix(f)
        self.g = it.g
        ctx, dt = it.x0.ctx, it.x0.dtype
        m = self.A.m
        x = it.x0.copy()
        self.Ax = self.A.mul(x)  # one pass
        self.u = HIPVector.empty(m, dt, ctx)  # grad loss(A x)
        self.Az = HIPVector.empty(m, dt, ctx)
        self.u_z = HIPVector.empty(m, dt, ctx)
        self.passes = 1
        f_x, _ = self.loss.value_and_gradient(self.Ax, out=self.u)
        if it.gamma is None:  # lower_bound_smoothness_constant (fb_tools.jl:7-19): init only, generic passes
            grad0 = self.A.mul_adjoint(self.u)
            gamma = R(R(1) / lower_bound_smoothness_constant(f, x, grad0))
            self.passes += 3
        else:
            gamma = R(it.gamma)
        s = state_cls(x=x, f_x=R(f_x), grad_f_x=x.similar(), gamma=gamma, y=x.similar(), z=x.similar(), res=x.similar())
        s.res_inf = None
        self.s = s
        self.stats = None
        self.sweep()  # grad_f_x, y, z, res, A z
        if with_prev:
            s.z_prev = x.copy()
            self.Az_prev = self.Ax.similar().copy_from(self.Ax)

    def sweep(self):
        s = self.s
        sc = self.A.fused_tn(self.u, s.x, s.gamma, self.g, s.grad_f_x, s.y, s.z, s.res, self.Az)
        self.passes += 1
        s.g_z, s.res_inf = sc[0], sc[1]
        self.stats = (sc[2], sc[3])  # <grad_f_x, res>, ||res||^2 of this pair
        self.it.counters["a_passes"] = self.passes

    def model(self, gamma):
        """f_model(f_x, grad_f_x, res, 1 / gamma)  (fb_tools.jl:3-5), from the sweep's reductions when they are current"""
        s, R = self.s, self.s.x.dtype.type
        if self.stats is not None:
            return R(R(s.f_x) - self.stats[0] + (R(1) / R(gamma) / R(2)) * self.stats[1])
        return f_model(s.f_x, s.grad_f_x, s.res, R(1) / R(gamma))

    def backtrack(self):
        """backtrack_stepsize!  (fb_tools.jl:24-63, alpha = 1) with f(z) = loss(A z): returns f(z); leaves
        grad loss(A z) in u_z and A z in Az for the accepted z"""
        it, s = self.it, self.s
        R = s.x.dtype.type
        eps = R(np.finfo(R).eps)
        gamma = R(s.gamma)
        f_z_upp = self.model(gamma)
        f_z, _ = self.loss.value_and_gradient(self.Az, out=self.u_z)
        tol = R(10) * eps * (R(1) + abs(f_z))
        nbt = 0
        while f_z > f_z_upp + tol and gamma >= it.minimum_gamma:
            gamma = R(gamma * it.reduce_gamma)
            s.y.axpby_(1.0, s.x, -gamma, s.grad_f_x)
            s.g_z = prox_(s.z, self.g, s.y, gamma)
            s.res.axpby_(1.0, s.x, -1.0, s.z)
            self.stats, s.res_inf = None, None
            f_z_upp = self.model(gamma)
            self.A.mul(s.z, self.Az)  # the pass a rejected trial costs
            self.passes += 1
            f_z, _ = self.loss.value_and_gradient(self.Az, out=self.u_z)
            tol = R(10) * eps * (R(1) + abs(f_z))
            nbt += 1
        if gamma < it.minimum_gamma:
            warnings.warn(f"stepsize `gamma` became too small ({gamma})")
        s.gamma = gamma
        s.n_backtracks = nbt
        it.counters["backtracks"] = it.counters.get("backtracks", 0) + nbt
        it.counters["a_passes"] = self.passes
        return R(f_z)


def iter_fb(it, state_cls):
    """forward_backward.jl:65-123 on f = Composed(loss, A)"""
    w = _Sweep(it, state_cls, with_prev=False)
    s = w.s
    R = s.x.dtype.type
    yield s
    while True:
        if it.adaptive:  # :90-110
            s.gamma = R(s.gamma * it.increase_gamma)
            s.f_x = w.backtrack()
            w.u, w.u_z = w.u_z, w.u  # grad loss at the new x = z
        s.x, s.z = s.z, s.x  # :109 / :112
        w.Ax, w.Az = w.Az, w.Ax
        if not it.adaptive:  # :113-114
            s.f_x = R(w.loss.value_and_gradient(w.Ax, out=w.u)[0])
        w.sweep()  # :117-120 (and A z for the next iteration)
        yield s


def iter_ffb(it, state_cls):
    """fast_forward_backward.jl:73-145 on f = Composed(loss, A)"""
    from .fast_forward_backward import call_extrapolate

    w = _Sweep(it, state_cls, with_prev=True)
    s = w.s
    R = s.x.dtype.type
    seq = iter(it.extrapolation_sequence) if it.extrapolation_sequence is not None else AdaptiveNesterovSequence(it.mf, R)
    s.extrapolation_sequence, s.beta = seq, R(0)
    yield s
    while True:
        if it.adaptive:  # :110-129
            s.gamma = R(s.gamma * it.increase_gamma)
            w.backtrack()
        elif it.gamma is not None:
            s.gamma = R(it.gamma)  # :131
        beta = seq.next(s.gamma) if isinstance(seq, AdaptiveNesterovSequence) else R(next(seq))  # :99-104
        s.beta = beta
        call_extrapolate(s.x, s.z, s.z_prev, beta)  # :135
        w.Ax.axpby_(float(R(1) + beta), w.Az, -float(beta), w.Az_prev)  # A x without a pass
        s.z_prev, s.z = s.z, s.z_prev  # :136
        w.Az_prev, w.Az = w.Az, w.Az_prev
        s.f_x = R(w.loss.value_and_gradient(w.Ax, out=w.u)[0])  # :138
        w.sweep()  # :138-142 (and A z for the next iteration)
        yield s


def try_iter(it, state_cls, fast):
    """the composed iteration, or None when the sweep kernel does not cover this matrix (too many rows, sharded)"""
    try:
        gen = (iter_ffb if fast else iter_fb)(it, state_cls)
        first = next(gen)
    except ProxGradError as e:
        if e.code == _lib.PG_ERR_UNSUPPORTED:  # fall back to the generic engine (any other failure propagates)
            return None
        raise

    def chain():
        yield first
        yield from gen

    return chain()
