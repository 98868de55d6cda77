"""ZeroFPR -- mirror of src/algorithms/zerofpr.jl: minimize f(A x) + g(x) with a quasi-Newton
direction on the forward-backward residual and a line search on the forward-backward envelope.  Same device
primitives as PANOC (panoc.py); every array statement is a kernel of libproxgrad_hip."""
import numpy as np

from . import _lib
from ._lib import ProxGradError
from .algorithm import IterativeAlgorithm
from .lbfgs import LBFGSOperator
from .operators import prox_
from .panoc import PANOCIteration, value_and_gradient_into


class ZeroFPRState:
    """zerofpr.jl:56-80"""

    pass


class ZeroFPRIteration(PANOCIteration):
    """zerofpr.jl:39-52 (same keyword constructor as PANOC), Base.iterate :85-111 / :142-220."""

    def _init(self):
        R = self.x0.dtype.type
        s = ZeroFPRState()
        s.x = self.x0.copy()  # :86
        s.Ax = self._mul_start(s.x)  # :87 (a start from zero reads nothing)
        s.grad_f_Ax = s.Ax.similar()
        s.f_Ax, _ = value_and_gradient_into(self.f, s.Ax, s.grad_f_Ax)  # :88
        if self.gamma is None:  # :89-92
            s.gamma = R(self.alpha / self._lower_bound_smoothness_constant(s.x, s.grad_f_Ax))
        else:
            s.gamma = R(self.gamma)
        s.Az_next = s.Ax.similar()
        s.Az_next_valid, s.Az_next_of, s.Az_next_is_res = False, None, False
        s.res_stats = s.res_inf = None
        start_fused = False
        if self._fused_tn:  # :93-96 in ONE read of A, which also leaves A xbar for the first iteration's :167 / fb_tools.jl:43
            s.At_grad_f_Ax, s.y, s.xbar, s.res = (s.x.similar() for _ in range(4))
            try:
                sc = self.A.fused_tn(s.grad_f_Ax, s.x, s.gamma, self.g, s.At_grad_f_Ax, s.y, s.xbar, s.res, s.Az_next)
                self.counters["A_passes"] += 1
                s.g_xbar = sc[0]
                s.Az_next_valid, s.Az_next_of = True, s.xbar
                s.res_stats, s.res_inf = (sc[1], sc[2], sc[3]), sc[1]
                start_fused = True
            except ProxGradError as e:
                if e.code != _lib.PG_ERR_UNSUPPORTED:
                    raise
                self._fused_tn = False
        if not start_fused:
            s.At_grad_f_Ax = self._mul_adj(None, s.grad_f_Ax)  # :93
            s.y = s.x.similar().axpby_(1.0, s.x, -s.gamma, s.At_grad_f_Ax)  # :94
            s.xbar = s.x.similar()
            s.g_xbar = prox_(s.xbar, self.g, s.y, s.gamma)  # :95
            s.res = s.x.similar().axpby_(1.0, s.x, -1.0, s.xbar)
        s.H = self.directions.initialize(s.x)
        s.tau = R(0)
        for name in ("At_grad_f_Axbar", "xbarbar", "res_xbar", "xbar_prev", "res_xbar_prev", "d"):
            setattr(s, name, s.x.similar())
        for name in ("Axbar", "grad_f_Axbar", "Ad"):
            setattr(s, name, s.Ax.similar())
        # Two trial points of the line search per sweep (pg_mat_fused_tn_pair): a second set of everything a trial point writes.
        # Allocated at the first use.
        s.pair, s.pair_sweeps = bool(self._fused_tn) and bool(getattr(self, "pair_trials", True)), 0
        s.trio, s.trio_sweeps = s.pair and bool(getattr(self, "trio_trials", True)), 0
        s.is_prev_set = False
        s.img = self._images and self._fused_tn and isinstance(s.H, LBFGSOperator)
        s.img_prev_set = False
        if s.img:  # the image slab (panoc.py): A d of :194 without reading A
            s.H.images_enable(s.Ax.n)
            for name in ("Ares", "Axbar_prev", "Ares_prev"):
                setattr(s, name, s.Ax.similar())
        return s

    def _step(self, s):
        R = s.x.dtype.type
        if self.adaptive:  # :143-165
            gamma_prev = s.gamma
            s.gamma, s.g_xbar, f_Axbar, f_Axbar_upp = self._backtrack_stepsize(s, s.xbar, s.g_xbar, s.Axbar, s.grad_f_Axbar)
            if s.gamma != gamma_prev and s.H is not None:
                s.H.reset_()
        else:  # :166-171
            if s.Az_next_valid and s.Az_next_of is s.xbar:
                s.Axbar.copy_from(s.Az_next)  # the last sweep of the previous iteration already formed A xbar
                s.Az_next_valid = False
            else:
                self._mul(s.Axbar, s.xbar)  # :167
            f_Axbar, _ = value_and_gradient_into(self.f, s.Axbar, s.grad_f_Axbar)
            f_Axbar_upp = self._model(s)
        FBE_x = R(f_Axbar_upp + s.g_xbar)  # :174
        sigma = R(self.beta * (R(0.5) / s.gamma) * (R(1) - self.alpha))  # :196
        tol = R(10) * R(np.finfo(R).eps) * (R(1) + abs(FBE_x))
        threshold = R(FBE_x - sigma * self._res_sq(s) + tol)  # :198 (moved up: state.res / its reductions are not touched before)
        # :177-180.  With the image slab on, as ONE sweep that also leaves A res_xbar = A (xbar - xbarbar)
        # (pg_mat_fused_tn_res): the image of the direction (:194) and the images of the next pair (:118-126) then come
        # from products of the residuals themselves and from A xbar, a product as well (the second sweep leaves it) --
        # nothing here is a running sum, so nothing drifts.
        s.img = s.img and self._fused_tn
        fused_res = False
        if s.img:
            try:
                self.A.fused_tn(s.grad_f_Axbar, s.xbar, s.gamma, self.g, s.At_grad_f_Axbar, s.y, s.xbarbar, s.res_xbar, s.Ares,
                                image_of_res=True)
                self.counters["A_passes"] += 1
                fused_res = True
            except ProxGradError as e:
                if e.code != _lib.PG_ERR_UNSUPPORTED:
                    raise
                self._fused_tn = s.img = False
        if not fused_res:
            self._mul_adj(s.At_grad_f_Axbar, s.grad_f_Axbar)  # :177
            s.y.axpby_(1.0, s.xbar, -s.gamma, s.At_grad_f_Axbar)  # :178
            prox_(s.xbarbar, self.g, s.y, s.gamma)  # :179
            s.res_xbar.axpby_(1.0, s.xbar, -1.0, s.xbarbar)  # :180
        if s.is_prev_set and s.H is not None:  # :182-184 (update_direction_state! :118-126)
            s.xbar_prev.axpby_(1.0, s.xbar, -1.0, s.xbar_prev)
            s.res_xbar_prev.axpby_(1.0, s.res_xbar, -1.0, s.res_xbar_prev)
            s.H.update_(s.xbar_prev, s.res_xbar_prev)
            if fused_res and s.img_prev_set:
                s.Axbar_prev.axpby_(1.0, s.Axbar, -1.0, s.Axbar_prev)
                s.Ares_prev.axpby_(1.0, s.Ares, -1.0, s.Ares_prev)
                s.H.images_update_(s.Axbar_prev, s.Ares_prev)
        s.xbar_prev.copy_from(s.xbar)  # :186-188
        s.res_xbar_prev.copy_from(s.res_xbar)
        s.is_prev_set = True
        s.img_prev_set = fused_res
        if fused_res:
            s.Axbar_prev.copy_from(s.Axbar)
            s.Ares_prev.copy_from(s.Ares)
        use_img = fused_res and s.H.images_ready()
        if s.H is not None:  # :190 (set_next_direction! :113-116)
            s.H.mul_(s.d, s.res_xbar)
            s.d.axpby_(-1.0, s.d)
        else:
            s.d.axpby_(-1.0, s.res)
        s.tau = R(1)  # :193
        if use_img:  # :194 without reading A: d = -(H res_xbar)
            s.H.images_mul_(s.Ad, s.Ares)
            s.Ad.axpby_(-1.0, s.Ad)
        elif fused_res and s.H is not None and getattr(s.H, "updates_since_reset", 1) == 0:
            # an EMPTY memory (the first iteration, and the one after every reset!): mul! is the identity (lbfgs.jl:64-71 with
            # currmem = 0, H = 1), d = -res_xbar, and A d = -A res_xbar is what the sweep above just left
            s.Ad.axpby_(-1.0, s.Ares)
        else:
            self._mul(s.Ad, s.d)  # :194
        # :200-217.  Every trial point tau is a sweep of its own in the reference (A' grad f(A x), the forward-backward step; here
        # also A xbar for the next iteration).  Here a sweep carries the trial points of tau, tau / 2 AND tau / 4 (three r slices,
        # three accumulator sets on one register tile, pg_mat_fused_tn_trio; two where only the pair sweep applies): rejected
        # trials cost no further read of A until the points carried are used up.  The decisions are the reference's -- a point
        # evaluated ahead is only looked at after the one before it was rejected.
        TRIAL = ("x", "Ax", "grad_f_Ax", "At_grad_f_Ax", "y", "xbar", "res", "Az_next")
        spec = []  # (tau, f_Ax, scalars, suffix of the buffer set) of the trial points the last sweep evaluated ahead
        for k in range(1, self.max_backtracks + 1):
            tau_next = R(0) if k >= self.max_backtracks - 1 else R(s.tau / R(2))  # :216
            fused = False
            if spec and spec[0][0] == s.tau:  # evaluated ahead: that buffer set becomes the state, no sweep
                _, s.f_Ax, sc, sfx = spec.pop(0)
                for name in TRIAL:
                    a_, b_ = getattr(s, name), getattr(s, name + sfx)
                    setattr(s, name, b_), setattr(s, name + sfx, a_)
                fused = True
            else:
                spec = []
                s.x.axpby_(1.0, s.xbar_prev, s.tau, s.d)  # :201
                s.Ax.axpby_(1.0, s.Axbar, s.tau, s.Ad)  # :202
                s.f_Ax, _ = value_and_gradient_into(self.f, s.Ax, s.grad_f_Ax)  # :204-205
                # (measured at config 4's size, profiles/r5_pair_sweep_rate.log: a pair sweep and a three-point sweep both cost 1.01-1.03
                # single sweeps, so further points are carried whenever there are any)
                ahead = []  # the trial points behind this one: (tau, buffer set)
                if self._fused_tn and s.pair and tau_next > 0:
                    ahead.append((tau_next, "_sp"))
                    tau_next2 = R(0) if k + 1 >= self.max_backtracks - 1 else R(tau_next / R(2))
                    if s.trio and tau_next2 > 0:
                        ahead.append((tau_next2, "_sp2"))
                f_ahead = []
                for t_, sfx in ahead:
                    if not hasattr(s, "x" + sfx):
                        for name in TRIAL:
                            setattr(s, name + sfx, getattr(s, name).similar())
                    getattr(s, "x" + sfx).axpby_(1.0, s.xbar_prev, t_, s.d)
                    getattr(s, "Ax" + sfx).axpby_(1.0, s.Axbar, t_, s.Ad)
                    f_ahead.append(value_and_gradient_into(self.f, getattr(s, "Ax" + sfx), getattr(s, "grad_f_Ax" + sfx))[0])
                outs = lambda sfx: tuple(getattr(s, n + sfx) for n in ("At_grad_f_Ax", "y", "xbar", "res", "Az_next"))  # noqa: E731
                if len(ahead) == 2:
                    try:
                        scs = self.A.fused_tn_trio((s.grad_f_Ax, s.grad_f_Ax_sp, s.grad_f_Ax_sp2), (s.x, s.x_sp, s.x_sp2), s.gamma, self.g,
                                                   (outs(""), outs("_sp"), outs("_sp2")))
                        sc = scs[0]
                        spec = [(ahead[0][0], f_ahead[0], scs[1], "_sp"), (ahead[1][0], f_ahead[1], scs[2], "_sp2")]
                        s.trio_sweeps += 1
                        fused = True
                    except ProxGradError as e:
                        if e.code != _lib.PG_ERR_UNSUPPORTED:
                            raise
                        s.trio = False  # no three-point sweep for this column length: two points
                        ahead = ahead[:1]
                if not fused and len(ahead) == 1:
                    try:
                        sc, sc2 = self.A.fused_tn_pair(s.grad_f_Ax, s.x, s.grad_f_Ax_sp, s.x_sp, s.gamma, self.g, outs(""), outs("_sp"))
                        spec = [(ahead[0][0], f_ahead[0], sc2, "_sp")]
                        s.pair_sweeps += 1
                        fused = True
                    except ProxGradError as e:
                        if e.code != _lib.PG_ERR_UNSUPPORTED:
                            raise
                        s.pair = False  # this column length has no two-point sweep: one trial point per sweep
                if not fused and self._fused_tn:  # :206-209 and the A xbar of the next iteration (fb_tools.jl:43 / :167) in one read of A
                    try:
                        sc = self.A.fused_tn(s.grad_f_Ax, s.x, s.gamma, self.g, s.At_grad_f_Ax, s.y, s.xbar, s.res, s.Az_next)
                        fused = True
                    except ProxGradError as e:
                        if e.code != _lib.PG_ERR_UNSUPPORTED:
                            raise
                        self._fused_tn = False
                if fused:
                    self.counters["A_passes"] += 1
            if fused:
                s.g_xbar = sc[0]
                s.Az_next_valid, s.Az_next_of = True, s.xbar
                s.res_stats, s.res_inf = (sc[1], sc[2], sc[3]), sc[1]  # the sweep's own reductions of (At_grad, res)
            else:
                s.res_stats = s.res_inf = None
                self._mul_adj(s.At_grad_f_Ax, s.grad_f_Ax)  # :206
                s.y.axpby_(1.0, s.x, -s.gamma, s.At_grad_f_Ax)  # :207
                s.g_xbar = prox_(s.xbar, self.g, s.y, s.gamma)  # :208
                s.res.axpby_(1.0, s.x, -1.0, s.xbar)  # :209
                s.Az_next_valid = False
            FBE_x = R(self._model(s) + s.g_xbar)  # :210
            if FBE_x <= threshold:
                break
            s.tau = tau_next  # :216
        return s


def default_stopping_criterion(tol, iteration, state):
    """zerofpr.jl:222-223"""
    R = state.res.dtype.type
    res_inf = state.res_inf if getattr(state, "res_inf", None) is not None else state.res.norm_inf()
    return R(res_inf) / R(state.gamma) <= R(tol)


def default_solution(iteration, state):
    """zerofpr.jl:224"""
    return state.xbar


def default_display(it, iteration, state):
    """zerofpr.jl:225-232"""
    print("%5d | %.3e | %.3e | %.3e" % (it, state.gamma, state.res.norm_inf() / state.gamma, state.tau))


def ZeroFPR(*, maxit=1_000, tol=1e-8, stop=None, solution=default_solution, verbose=False, freq=10,
            display=default_display, **kwargs):
    """zerofpr.jl:262-280"""
    if stop is None:
        stop = lambda iteration, state: default_stopping_criterion(tol, iteration, state)
    return IterativeAlgorithm(ZeroFPRIteration, maxit=maxit, stop=stop, solution=solution, verbose=verbose, freq=freq,
                              display=display, **kwargs)
