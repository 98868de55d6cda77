"""DouglasRachford splitting -- mirror of src/algorithms/douglas_rachford.jl (BASELINE config 3).

Engines: ``fused`` (f = SeparableQuadratic, g in {IndBox(scalars), NormL1, Zero}: the five array statements of
douglas_rachford.jl:58-62 run as ONE HBM sweep, pg_dr_step) and ``generic`` (any two operators with ``prox_``).
"""
import ctypes as C

from ._lib import call
from .algorithm import IterativeAlgorithm
from .device import as_hipvector
from .operators import IndBox, NormL1, SeparableQuadratic, Zero, prox_


class DouglasRachfordState:
    """douglas_rachford.jl:45-51"""

    def __init__(self, x):
        self.x = x
        self.y = x.similar()
        self.r = x.similar()
        self.z = x.similar()
        self.res = x.similar()
        self.res_inf = None
        self.f_y = None
        self.g_z = None


def _fused_ok(f, g):
    if not isinstance(f, SeparableQuadratic) or not isinstance(g, (IndBox, NormL1, Zero)):
        return False
    return getattr(g, "_scalar", True)  # per-element bounds / weights: the generic engine


class DouglasRachfordIteration:
    """douglas_rachford.jl:30-41 (keyword constructor: f, g, x0, gamma) and Base.iterate :53-63.
    ``materialize=False`` (fused engine only) skips writing r, z, res: x and y are the only n-vector writes."""

    def __init__(self, *, f=None, g=None, x0, gamma, engine=None, materialize=True, lookahead=True):
        self.f = f if f is not None else Zero()
        self.g = g if g is not None else Zero()
        self.x0 = as_hipvector(x0)
        self.gamma = self.x0.dtype.type(gamma)
        if engine is None:
            engine = "fused" if _fused_ok(self.f, self.g) else "generic"
        if engine == "fused" and not _fused_ok(self.f, self.g):
            raise TypeError("engine='fused' needs f = SeparableQuadratic and g in {IndBox(scalar bounds), NormL1(scalar lam), Zero}")
        self.engine = engine
        self.materialize = bool(materialize)
        # fused engine: iteration k + 1 is launched (into a second set of state vectors) before iteration k's scalars are read
        # (pg_dr_step_async): the state yielded is the reference's state k, bit for bit; its vectors stay valid until the NEXT
        # state is yielded, as in the reference (states are updated in place).  lookahead=False: one launch, one read-back.
        self.lookahead = bool(lookahead)

    def device_run(self, maxit, tol, block=64):
        """The driver loop of ProximalAlgorithms.jl:114-123 with the default stop rule, inside the library
        (pg_dr_run): ``block`` iterations per HBM sweep.  Returns ``(state, k)``; the state is bit-identical to the one
        the step-by-step loop stops at."""
        if self.engine != "fused":
            raise TypeError("device_run needs the fused engine")
        R = self.x0.dtype.type
        s = DouglasRachfordState(self.x0.copy())
        x_alt = s.x.similar()
        dv, d, qv, q = self.f.c_params()
        p0, p1 = self.g.g_params()
        sc = (C.c_double * 3)()
        k = C.c_int64()
        opt = (lambda v: v.vp) if self.materialize else (lambda v: None)
        call("pg_dr_run", s.x.ctx.handle, s.x.pg_dtype, s.x.n, s.x.vp, x_alt.vp, s.y.vp, opt(s.r), opt(s.z), opt(s.res),
             dv, d, qv, q, self.g.g_kind, p0, p1, float(self.gamma), float(tol), int(maxit), int(block), C.byref(k), sc)
        s.res_inf, s.f_y, s.g_z = R(sc[0]), R(sc[1]), R(sc[2])
        return s, int(k.value)

    def __iter__(self):
        R = self.x0.dtype.type
        s = DouglasRachfordState(self.x0.copy())  # state = DouglasRachfordState(x = copy(iter.x0))  (:55)
        if self.engine == "fused":
            dv, d, qv, q = self.f.c_params()
            p0, p1 = self.g.g_params()
            sc = (C.c_double * 3)()
            opt = (lambda v: v.vp) if self.materialize else (lambda v: None)
            if self.lookahead:
                h, dt, n, gk, gm = s.x.ctx.handle, s.x.pg_dtype, s.x.n, self.g.g_kind, float(self.gamma)
                names = ("x", "y", "r", "z", "res")
                sets = [{k: getattr(s, k) for k in names}, {k: s.x.similar() for k in names}]
                x_in = s.x.copy()  # (state.x is written by the first iteration: it reads a copy of x0)

                def launch(src, dst, slot):
                    call("pg_dr_step_async", h, dt, n, src.vp, dst["x"].vp, dst["y"].vp, opt(dst["r"]), opt(dst["z"]), opt(dst["res"]),
                         dv, d, qv, q, gk, p0, p1, gm, slot)

                launch(x_in, sets[0], 0)
                i = 0
                while True:
                    launch(sets[i]["x"], sets[1 - i], 1 - i)  # iteration k + 1, behind iteration k on the stream
                    call("pg_dr_step_wait", h, i, sc)  # iteration k (its successor keeps running)
                    for k in names:
                        setattr(s, k, sets[i][k])
                    s.res_inf, s.f_y, s.g_z = R(sc[0]), R(sc[1]), R(sc[2])
                    yield s
                    i = 1 - i
            while True:
                call("pg_dr_step", s.x.ctx.handle, s.x.pg_dtype, s.x.n, s.x.vp, s.y.vp, opt(s.r), opt(s.z), opt(s.res),
                     dv, d, qv, q, self.g.g_kind, p0, p1, float(self.gamma), sc)
                s.res_inf, s.f_y, s.g_z = R(sc[0]), R(sc[1]), R(sc[2])
                yield s
        else:
            while True:
                self.body(s)
                yield s

    @property
    def graph_safe(self):
        return self.engine == "generic"  # the fused engine already is one kernel per iteration / has its own loop

    def init_state(self):
        return DouglasRachfordState(self.x0.copy())

    def body(self, s):
        """generic engine: the five statements of douglas_rachford.jl:58-62 as library calls (allocation-free)"""
        prox_(s.y, self.f, s.x, self.gamma, want_value=False)  # :58 (the reference discards the values)
        s.r.axpby_(2.0, s.y, -1.0, s.x)  # :59
        prox_(s.z, self.g, s.r, self.gamma, want_value=False)  # :60
        s.res.axpby_(1.0, s.y, -1.0, s.z)  # :61
        s.x.axpby_(1.0, s.x, -1.0, s.res)  # :62
        s.res_inf = None


def default_stopping_criterion(tol, iteration, state):
    """norm(state.res, Inf) / iter.gamma <= tol   (douglas_rachford.jl:65-69)"""
    R = state.x.dtype.type
    res_inf = state.res_inf if state.res_inf is not None else state.res.norm_inf()
    return R(res_inf) / R(iteration.gamma) <= R(tol)


def default_solution(iteration, state):
    """douglas_rachford.jl:70"""
    return state.y


def default_display(it, iteration, state):
    """douglas_rachford.jl:71-72"""
    res_inf = state.res_inf if state.res_inf is not None else state.res.norm_inf()
    print("%5d | %.3e" % (it, res_inf / iteration.gamma))


def DouglasRachford(*, maxit=1_000, tol=1e-8, stop=None, solution=default_solution, verbose=False, freq=100,
                    display=default_display, device_loop=False, check_every=64, graph=False, **kwargs):
    """douglas_rachford.jl:101-119.  device_loop=True (default stop rule, fused engine): the driver loop runs inside
    the library, ``check_every`` (1, 8, 16, 32 or 64) iterations per HBM sweep; same iterates, same iteration count."""
    dl = (tol, int(check_every)) if (device_loop and stop is None) else None
    if stop is None:
        stop = lambda iteration, state: default_stopping_criterion(tol, iteration, state)
    return IterativeAlgorithm(DouglasRachfordIteration, maxit=maxit, stop=stop, solution=solution, verbose=verbose,
                              freq=freq, display=display, device_loop=dl, graph=graph, **kwargs)
