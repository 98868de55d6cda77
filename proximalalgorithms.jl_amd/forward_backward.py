"""ForwardBackward (proximal gradient) -- mirror of src/algorithms/forward_backward.jl.

Two engines behind one iterator type:
  * ``fused``   : f = LeastSquares, g in {NormL1, IndBox, Zero}: the whole iteration body runs inside
                  libproxgrad_hip (csrc/pg_iter.hip) -- one C call per iteration.
  * ``generic`` : any operator objects honouring value_and_gradient / prox_ (drop-in boundary #2); the body
                  below is the reference's, on device vectors.
"""
from .algorithm import IterativeAlgorithm
from .device import as_hipvector
from .fb_tools import backtrack_stepsize_, lower_bound_smoothness_constant
from .operators import Zero, fused_supported, prox_, value_and_gradient
from . import _composed, _lib
from ._composed import composed_supported
from ._fused import FusedIteration


class _LazyVectors:
    """State base: the vector fields of a fused iteration are views of library-owned memory whose pointers
    change when the library swaps references (x <-> z, ...); they are materialised on first access after a step
    (one C call) so that a plain `for state in iteration` loop pays nothing for them."""

    _vector_fields = ()

    def _bind(self, fused):
        object.__setattr__(self, "_fused", fused)
        object.__setattr__(self, "_views", None)

    def _invalidate(self):
        object.__setattr__(self, "_views", None)

    def __getattr__(self, name):  # only reached when normal lookup fails
        if name in type(self)._vector_fields:
            fused = self.__dict__.get("_fused")
            if fused is None:
                return None
            views = self.__dict__.get("_views")
            if views is None:
                views = fused.view()
                object.__setattr__(self, "_views", views)
            return views[name]
        raise AttributeError(name)


class ForwardBackwardState(_LazyVectors):
    """forward_backward.jl:52-63"""

    _vector_fields = ("x", "grad_f_x", "y", "z", "res", "grad_f_z")

    def __init__(self, **kw):
        self.f_x = self.gamma = self.g_z = self.res_inf = None
        self.n_backtracks = 0
        for k, v in kw.items():
            setattr(self, k, v)


class ForwardBackwardIteration:
    """forward_backward.jl:38-48 (keyword constructor), Base.iterate :65-84 / :86-123.

    Iterating yields the (mutated in place) state object forever (``IteratorSize = IsInfinite``)."""

    def __init__(self, *, f=None, g=None, x0, Lf=None, gamma=None, adaptive=None, minimum_gamma=1e-7,
                 reduce_gamma=0.5, increase_gamma=1.0, engine=None, single_sweep=True):
        self.single_sweep = bool(single_sweep)  # fused engine: one read of A per iteration where possible
        self.f = f if f is not None else Zero()
        self.g = g if g is not None else Zero()
        ctx = getattr(self.f, "ctx", None)
        self.x0 = as_hipvector(x0, ctx)
        R = self.x0.dtype.type
        self.Lf = Lf
        self.gamma = gamma if gamma is not None else (None if Lf is None else R(1) / R(Lf))  # :43
        self.adaptive = (self.gamma is None) if adaptive is None else bool(adaptive)  # :44
        self.minimum_gamma = R(minimum_gamma)
        self.reduce_gamma = R(reduce_gamma)
        self.increase_gamma = R(increase_gamma)
        auto = engine is None
        if engine is None:
            engine = "fused" if fused_supported(self.f, self.g) else (
                "composed" if (composed_supported(self.f, self.g) and single_sweep) else "generic")
        # f = LeastSquares with the adaptive step: the library's fused iteration needs two sweeps there (the accepted
        # trial's gradient is a second product), the composed re-association one -- taken once A is large enough for the
        # sweeps to dominate the extra host calls (64 MiB)
        if (auto and engine == "fused" and self.adaptive and single_sweep and getattr(self.f, "comm", 1) is None
                and self.f.A.m * self.f.A.n * self.f.A.dtype.itemsize >= (64 << 20) and composed_supported(self.f, self.g)):
            engine = "composed"
        if engine == "composed" and not composed_supported(self.f, self.g):
            raise TypeError("engine='composed' needs f = Composed(loss, device matrix) or an unsharded LeastSquares, and g in "
                            "{NormL1, IndBox(scalar bounds), Zero}")
        if engine == "fused" and not fused_supported(self.f, self.g):
            raise TypeError("engine='fused' needs f = LeastSquares and g in {NormL1, IndBox, Zero}")
        self.engine = engine
        self.counters = {}

    # ---- fused engine ----
    def _iter_fused(self, resume_blob=None):
        R = self.x0.dtype.type
        fi = FusedIteration(self.f, self.g, fast=False, Lf=self.Lf, gamma=self.gamma, adaptive=self.adaptive,
                            minimum_gamma=self.minimum_gamma, reduce_gamma=self.reduce_gamma,
                            increase_gamma=self.increase_gamma, single_sweep=self.single_sweep)
        self._fused = fi
        state = ForwardBackwardState()

        state._bind(fi)

        def refresh(sc):
            state._invalidate()
            state.f_x, state.gamma, state.g_z = R(sc.f_x), R(sc.gamma), R(sc.g_z)
            state.res_inf = R(sc.res_inf)
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
            refresh(fi.state_upload(resume_blob))
        while True:
            refresh(fi.step())
            yield state

    # ---- generic engine: the reference body on device vectors ----
    def _iter_generic(self):
        R = self.x0.dtype.type
        x = self.x0.copy()  # :66
        f_x, grad_f_x = value_and_gradient(self.f, x)  # :67
        if self.gamma is None:  # :68-70
            gamma = R(R(1) / lower_bound_smoothness_constant(self.f, x, grad_f_x))
        else:
            gamma = R(self.gamma)
        y = x.similar().axpby_(1.0, x, -gamma, grad_f_x)  # :71
        z = x.similar()
        g_z = prox_(z, self.g, y, gamma)  # :72
        res = x.similar().axpby_(1.0, x, -1.0, z)
        s = ForwardBackwardState(x=x, f_x=R(f_x), grad_f_x=grad_f_x, gamma=gamma, y=y, z=z, g_z=g_z, res=res,
                                 grad_f_z=x.similar())
        s.res_inf = None
        yield s
        while True:
            if self.adaptive:  # :90-110
                s.gamma = R(s.gamma * self.increase_gamma)
                s.gamma, s.g_z, s.f_x, _ = backtrack_stepsize_(
                    s.gamma, self.f, self.g, s.x, s.f_x, s.grad_f_x, s.y, s.z, s.g_z, s.res, s.grad_f_z,
                    minimum_gamma=self.minimum_gamma, reduce_gamma=self.reduce_gamma, counters=self.counters)
                s.x, s.z = s.z, s.x
                s.grad_f_x, s.grad_f_z = s.grad_f_z, s.grad_f_x
            else:  # :111-115
                s.x, s.z = s.z, s.x
                s.f_x, grad = value_and_gradient(self.f, s.x)
                s.grad_f_x.copy_from(grad)
            s.y.axpby_(1.0, s.x, -s.gamma, s.grad_f_x)  # :117
            s.g_z = prox_(s.z, self.g, s.y, s.gamma)  # :118
            s.res.axpby_(1.0, s.x, -1.0, s.z)  # :120
            s.res_inf = None
            yield s

    def save_state(self):
        """All algorithm memory of the running solve as bytes -- the state struct of the reference (forward_backward.jl:52-63), which
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
            gen = _composed.try_iter(self, ForwardBackwardState, fast=False)
            if gen is not None:
                return gen
            self.engine = "generic"  # the sweep kernel does not cover this matrix
        return self._iter_generic()


def _res_inf(state):
    return state.res_inf if state.res_inf is not None else state.res.norm_inf()


def default_stopping_criterion(tol, iteration, state):
    """norm(state.res, Inf) / state.gamma <= tol   (forward_backward.jl:125-126)"""
    R = state.res.dtype.type
    return R(_res_inf(state)) / R(state.gamma) <= R(tol)


def default_solution(iteration, state):
    """forward_backward.jl:127 -- state.z (aliased, not copied)"""
    return state.z


def default_display(it, iteration, state):
    """forward_backward.jl:128-129"""
    print("%5d | %.3e | %.3e" % (it, state.gamma, _res_inf(state) / state.gamma))


def ForwardBackward(*, maxit=10_000, tol=1e-8, stop=None, solution=default_solution, verbose=False, freq=100,
                    display=default_display, device_loop=False, check_every=1, **kwargs):
    """forward_backward.jl:161-179"""
    dl = (tol, int(check_every)) if (device_loop and stop is None and solution is default_solution) else None
    if stop is None:
        stop = lambda iteration, state: default_stopping_criterion(tol, iteration, state)
    return IterativeAlgorithm(ForwardBackwardIteration, maxit=maxit, stop=stop, solution=solution, verbose=verbose, device_loop=dl,
                              freq=freq, display=display, **kwargs)


# Aliases (forward_backward.jl:183-184)
ProximalGradientIteration = ForwardBackwardIteration
ProximalGradient = ForwardBackward
