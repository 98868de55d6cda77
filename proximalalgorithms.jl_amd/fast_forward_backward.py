"""FastForwardBackward (accelerated proximal gradient / FISTA) -- mirror of
src/algorithms/fast_forward_backward.jl.  Engines as in forward_backward.py."""
import copy
import itertools

import numpy as np

from . import _lib
from .algorithm import IterativeAlgorithm
from .device import as_hipvector
from .fb_tools import backtrack_stepsize_, lower_bound_smoothness_constant
from .forward_backward import _LazyVectors, _res_inf
from .nesterov import (AdaptiveNesterovSequence, ConstantNesterovSequence, FixedNesterovSequence,
                       SimpleNesterovSequence)
from .operators import Zero, fused_supported, prox_, value_and_gradient
from . import _composed
from ._composed import composed_supported
from ._fused import FusedIteration


class FastForwardBackwardState(_LazyVectors):
    """fast_forward_backward.jl:60-71"""

    _vector_fields = ("x", "grad_f_x", "y", "z", "res", "z_prev")

    def __init__(self, **kw):
        self.f_x = self.gamma = self.g_z = self.res_inf = self.beta = self.extrapolation_sequence = None
        self.n_backtracks = 0
        for k, v in kw.items():
            setattr(self, k, v)


class FastForwardBackwardIteration:
    """fast_forward_backward.jl:44-56 (keyword constructor), Base.iterate :73-97 / :106-145."""

    def __init__(self, *, f=None, g=None, x0, mf=0.0, Lf=None, gamma=None, adaptive=None, minimum_gamma=1e-7,
                 reduce_gamma=0.5, increase_gamma=1.0, extrapolation_sequence=None, engine=None, reuse_residual=True,
                 single_sweep=True):
        self.f = f if f is not None else Zero()
        self.g = g if g is not None else Zero()
        ctx = getattr(self.f, "ctx", None)
        self.x0 = as_hipvector(x0, ctx)
        R = self.x0.dtype.type
        self.mf = R(mf)
        self.Lf = Lf
        self.gamma = gamma if gamma is not None else (None if Lf is None else R(1) / R(Lf))  # :50
        self.adaptive = (self.gamma is None) if adaptive is None else bool(adaptive)  # :51
        self.minimum_gamma = R(minimum_gamma)
        self.reduce_gamma = R(reduce_gamma)
        self.increase_gamma = R(increase_gamma)
        self.extrapolation_sequence = extrapolation_sequence
        # fused engine, adaptive step: build A x - b from the residuals the line search holds (2 passes/iter, not 3)
        self.reuse_residual = bool(reuse_residual)
        # fused engine: ONE read of A per iteration where the operator allows it (pg_ls_fused_pass); False = two sweeps
        self.single_sweep = bool(single_sweep)
        if engine is None:
            engine = "fused" if fused_supported(self.f, self.g) else (
                "composed" if (composed_supported(self.f, self.g) and self.single_sweep) else "generic")
        if engine == "composed" and not composed_supported(self.f, self.g):
            raise TypeError("engine='composed' needs f = Composed(loss, device matrix) and g in {NormL1, IndBox(scalar bounds), Zero}")
        if engine == "fused" and not fused_supported(self.f, self.g):
            raise TypeError("engine='fused' needs f = LeastSquares and g in {NormL1, IndBox, Zero}")
        self.engine = engine
        self.counters = {}

    def _seq_spec(self):
        """Map the extrapolation sequence to the library's native kinds (else feed coefficients from the host)."""
        s = self.extrapolation_sequence
        R = self.x0.dtype
        if s is None:
            return _lib.PG_SEQ_ADAPTIVE, 0.0, 0.0, None
        if isinstance(s, FixedNesterovSequence) and np.dtype(s.R) == R:
            return _lib.PG_SEQ_FIXED, 0.0, 0.0, None
        if isinstance(s, SimpleNesterovSequence) and np.dtype(s.R) == R:
            return _lib.PG_SEQ_SIMPLE, 0.0, 0.0, None
        if isinstance(s, ConstantNesterovSequence) and np.dtype(s.R) == R:
            return _lib.PG_SEQ_CONSTANT, float(s.m), float(s.stepsize), None
        if isinstance(s, itertools.repeat):  # Iterators.repeated(beta): the library repeats the value itself
            try:  # only the INFINITE form (a finite repeat(beta, times) is exhausted after `times` draws, like the
                s.__length_hint__()  # reference's Iterators.Stateful): asking it for a length raises TypeError
                infinite = False
            except TypeError:
                infinite = True
            if infinite:
                beta = next(copy.copy(s))  # the caller's iterator is left untouched
                if isinstance(beta, (float, np.floating)) and float(R.type(beta)) == float(beta):
                    return _lib.PG_SEQ_REPEATED, float(beta), 0.0, None
        return _lib.PG_SEQ_HOST, 0.0, 0.0, iter(s)  # Iterators.Stateful(seq)  (:90-92)

    def _iter_fused(self, resume_blob=None):
        R = self.x0.dtype.type
        kind, p0, p1, host_iter = self._seq_spec()
        fi = FusedIteration(self.f, self.g, fast=True, Lf=self.Lf, gamma=self.gamma, adaptive=self.adaptive,
                            minimum_gamma=self.minimum_gamma, reduce_gamma=self.reduce_gamma,
                            increase_gamma=self.increase_gamma, mf=self.mf, seq_kind=kind, seq_p0=p0, seq_p1=p1,
                            reuse_residual=self.reuse_residual, single_sweep=self.single_sweep)
        self._fused = fi
        state = FastForwardBackwardState(extrapolation_sequence=self.extrapolation_sequence)

        state._bind(fi)

        def refresh(sc):
            state._invalidate()
            state.f_x, state.gamma, state.g_z = R(sc.f_x), R(sc.gamma), R(sc.g_z)
            state.res_inf, state.beta = R(sc.res_inf), R(sc.beta)
            state.n_backtracks = sc.n_backtracks
            self.counters["backtracks"] = self.counters.get("backtracks", 0) + sc.n_backtracks
            self.counters["a_passes"] = sc.a_passes
            state.flags = sc.flags
            if sc.flags & _lib.PG_FLAG_SWEEP_FALLBACK:  # this step's single sweep was lost and redone with two sweeps
                self.counters["sweep_fallbacks"] = self.counters.get("sweep_fallbacks", 0) + 1

        if resume_blob is None:
            refresh(fi.init(self.x0))
            yield state
        else:  # `iterate(iter, saved_state)`: the first state yielded is the one AFTER the saved one
            if host_iter is not None:
                raise ValueError("a host-drawn extrapolation sequence is not part of the saved state")
            refresh(fi.state_upload(resume_blob))
        while True:
            beta = float(next(host_iter)) if host_iter is not None else 0.0
            refresh(fi.step(beta))
            yield state

    def _iter_generic(self):
        R = self.x0.dtype.type
        x = self.x0.copy()  # :74
        f_x, grad_f_x = value_and_gradient(self.f, x)  # :75
        if self.gamma is None:  # :76-78
            gamma = R(R(1) / lower_bound_smoothness_constant(self.f, x, grad_f_x))
        else:
            gamma = R(self.gamma)
        y = x.similar().axpby_(1.0, x, -gamma, grad_f_x)  # :79
        z = x.similar()
        g_z = prox_(z, self.g, y, gamma)  # :80
        res = x.similar().axpby_(1.0, x, -1.0, z)  # :89
        if self.extrapolation_sequence is not None:  # :90-94
            seq = iter(self.extrapolation_sequence)
        else:
            seq = AdaptiveNesterovSequence(self.mf, R)
        s = FastForwardBackwardState(x=x, f_x=R(f_x), grad_f_x=grad_f_x, gamma=gamma, y=y, z=z, g_z=g_z, res=res,
                                     z_prev=x.copy(), extrapolation_sequence=seq, beta=R(0))
        yield s
        while True:
            if self.adaptive:  # :110-129
                s.gamma = R(s.gamma * self.increase_gamma)
                s.gamma, s.g_z, _, _ = backtrack_stepsize_(
                    s.gamma, self.f, self.g, s.x, s.f_x, s.grad_f_x, s.y, s.z, s.g_z, s.res, None,
                    minimum_gamma=self.minimum_gamma, reduce_gamma=self.reduce_gamma, counters=self.counters)
            else:
                s.gamma = R(self.gamma) if self.gamma is not None else s.gamma  # :131
            if isinstance(seq, AdaptiveNesterovSequence):  # :99-104
                beta = seq.next(s.gamma)
            else:
                beta = R(next(seq))
            s.beta = beta
            call_extrapolate(s.x, s.z, s.z_prev, beta)  # :135
            s.z_prev, s.z = s.z, s.z_prev  # :136
            s.f_x, grad = value_and_gradient(self.f, s.x)  # :138
            s.grad_f_x.copy_from(grad)  # :139
            s.y.axpby_(1.0, s.x, -s.gamma, s.grad_f_x)  # :140
            s.g_z = prox_(s.z, self.g, s.y, s.gamma)  # :141
            s.res.axpby_(1.0, s.x, -1.0, s.z)  # :142
            s.res_inf = None
            yield s

    def save_state(self):
        """All algorithm memory of the running solve as bytes -- the state struct of the reference (fast_forward_backward.jl:60-71), which
        `iterate(iter, state)` resumes from: state vectors, residuals, gamma, f_x, g_z, the extrapolation sequence's state.
        Engine "fused" only (pg_iter_state_download)."""
        if getattr(self, "_fused", None) is None:
            raise ValueError("save_state needs a started iteration on the fused engine (this one: engine=%r; pass engine='fused' to "
                             "the constructor -- the automatic choice takes the composed engine for an adaptive ForwardBackward on a "
                             "large unsharded matrix)" % (self.engine,))
        return self._fused.state_download()

    def resume(self, blob):
        """`Base.iterate(iter, saved_state)` over and over: an iterator of the states that FOLLOW the saved one, bit-identical
        to the solve the blob was taken from (a fresh library iterator with this iteration's options takes the blob:
        pg_iter_state_upload).  x0 is not read."""
        if self.engine != "fused":
            raise ValueError("resume needs the fused engine")
        return self._iter_fused(resume_blob=blob)

    def __iter__(self):
        if self.engine == "fused":
            return self._iter_fused()
        if self.engine == "composed":  # x -> loss(A x): one read of A per iteration (_composed.py)
            gen = _composed.try_iter(self, FastForwardBackwardState, fast=True)
            if gen is not None:
                return gen
            self.engine = "generic"  # the sweep kernel does not cover this matrix
        return self._iter_generic()


def call_extrapolate(x, z, z_prev, beta):
    """x .= z .+ beta .* (z .- z_prev)"""
    _lib.call("pg_extrapolate", x.ctx.handle, x.pg_dtype, x.n, x.vp, z.vp, z_prev.vp, float(beta))


def default_stopping_criterion(tol, iteration, state):
    """fast_forward_backward.jl:147-151"""
    R = state.res.dtype.type
    return R(_res_inf(state)) / R(state.gamma) <= R(tol)


def default_solution(iteration, state):
    """fast_forward_backward.jl:152"""
    return state.z


def default_display(it, iteration, state):
    """fast_forward_backward.jl:153-154"""
    print("%5d | %.3e | %.3e" % (it, state.gamma, _res_inf(state) / state.gamma))


def FastForwardBackward(*, maxit=10_000, tol=1e-8, stop=None, solution=default_solution, verbose=False, freq=100,
                        display=default_display, device_loop=False, check_every=1, **kwargs):
    """fast_forward_backward.jl:186-204.  device_loop=True (default stop/solution only): the driver loop runs inside
    the library -- one kernel launch for launch-bound sizes, else the in-library loop (batched by `check_every` when
    the step is fixed)."""
    dl = (tol, int(check_every)) if (device_loop and stop is None and solution is default_solution) else None
    if stop is None:
        stop = lambda iteration, state: default_stopping_criterion(tol, iteration, state)
    return IterativeAlgorithm(FastForwardBackwardIteration, maxit=maxit, stop=stop, solution=solution,
                              verbose=verbose, freq=freq, display=display, device_loop=dl, **kwargs)


# Aliases (fast_forward_backward.jl:208-209)
FastProximalGradientIteration = FastForwardBackwardIteration
FastProximalGradient = FastForwardBackward
