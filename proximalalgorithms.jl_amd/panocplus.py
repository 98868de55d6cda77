"""PANOCplus -- mirror of src/algorithms/panocplus.jl.  Same device primitives as PANOC."""
import warnings

import numpy as np

from . import _lib
from ._lib import ProxGradError
from .algorithm import IterativeAlgorithm
from .lbfgs import LBFGSOperator
from .operators import prox_
from .panoc import PANOCIteration, value_and_gradient_into


class PANOCplusState:
    """panocplus.jl:56-75"""

    pass


class PANOCplusIteration(PANOCIteration):
    """panocplus.jl:39-52 (same keyword constructor as PANOC), Base.iterate :85-128 / :168-240."""

    def _init(self):
        R = self.x0.dtype.type
        s = PANOCplusState()
        s.x = self.x0.copy()  # :86
        s.Ax = self._mul_start(s.x)
        s.grad_f_Ax = s.Ax.similar()
        s.f_Ax, _ = value_and_gradient_into(self.f, s.Ax, s.grad_f_Ax)
        if self.gamma is None:
            s.gamma = R(self.alpha / self._lower_bound_smoothness_constant(s.x, s.grad_f_Ax))
        else:
            s.gamma = R(self.gamma)
        s.At_grad_f_Ax = self._mul_adj(None, s.grad_f_Ax)
        s.y = s.x.similar().axpby_(1.0, s.x, -s.gamma, s.At_grad_f_Ax)
        s.z = s.x.similar()
        s.g_z = prox_(s.z, self.g, s.y, s.gamma)
        s.res = s.x.similar().axpby_(1.0, s.x, -1.0, s.z)
        s.H = self.directions.initialize(s.x)
        s.tau = R(0)
        for name in ("x_prev", "res_prev", "d", "At_grad_f_Az"):
            setattr(s, name, s.x.similar())
        for name in ("Az", "grad_f_Az"):
            setattr(s, name, s.Ax.similar())
        if self.gamma is None or self.adaptive:  # :105-121
            s.gamma, s.g_z, _, _ = self._backtrack_stepsize(s, s.z, s.g_z, s.Az, s.grad_f_Az)
        else:  # :122-126
            self._mul(s.Az, s.z)
            value_and_gradient_into(self.f, s.Az, s.grad_f_Az)
        self._mul_adj(s.At_grad_f_Az, s.grad_f_Az)  # :127
        # the image slab (panoc.py): x = x_prev + d with d = -H res_prev, so `mul!(state.Ax, iter.A, state.x)` (:199)
        # becomes A x_prev + (image of A (-res_prev)), A res_prev = A x_prev - A z_prev being held (:199, :210)
        s.img = self._images and isinstance(s.H, LBFGSOperator)
        s.img_steps = 0
        s.sp = None  # the next iteration's first pass, when the previous step took it ahead (_speculate)
        if s.img:
            s.H.images_enable(s.Ax.n)
            for name in ("Ax_prev", "Ad", "Ares", "Ares_prev", "As", "Ay"):
                setattr(s, name, s.Ax.similar())
            s.Ares.axpby_(1.0, s.Ax, -1.0, s.Az)  # A res at the initial point
        return s

    # what a first pass of the inner loop (direction, x = x_prev + d, its images, the sweep) writes: these are what the speculative
    # half iteration keeps in a second set of buffers until the next step takes them
    SPEC = ("d", "Ad", "x", "Ax", "grad_f_Ax", "At_grad_f_Ax", "y", "z", "res", "Ares", "Az")

    def _speculate(self, s):
        """`mul!(state.At_grad_f_Az, adjoint(iter.A), state.grad_f_Az)` (:225) is needed by the stopping criterion alone (:243) and
        nothing else depends on it -- but it is a read of A.  The FIRST pass of the next iteration's loop (:185-210 at tau = 1:
        direction from the memory updated at :237, x = x_prev + d, A x from the image slab, A' grad f(A x), the forward-backward
        step and A res) depends on nothing this one does not have either.  So both go through ONE sweep (pg_mat_fused_tn_pair_res:
        two r slices, two accumulator sets on one register tile; 1.02 single sweeps): this iteration's A' grad f(A z), and the next
        iteration's first pass into a second set of buffers, which the next step takes if nothing has changed -- one read of A
        per iteration instead of two.  The state the caller sees after this iteration is this iteration's.  Returns False (nothing
        done) where the pair sweep, the image slab or a complete set of images is not available: the plain product follows."""
        R = s.x.dtype.type
        if not (self._fused_tn and self.speculate and s.img and s.H is not None and s.H.images_ready()):
            return False
        if self.refresh_every > 0 and (s.img_steps + 1) % self.refresh_every == 0:
            return False  # the next pass is due a product A x: not from the slab
        if not hasattr(s, "sp_x"):
            for name in self.SPEC:
                setattr(s, "sp_" + name, getattr(s, name).similar())
            s.sp_junk_n = [s.x.similar() for _ in range(3)]
            s.sp_junk_m = s.Ax.similar()
        s.H.mul_(s.sp_d, s.res)  # set_next_direction! (:130-138) of the NEXT iteration: res becomes its res_prev
        s.sp_d.axpby_(-1.0, s.sp_d)
        s.H.images_mul_(s.sp_Ad, s.Ares)
        s.sp_Ad.axpby_(-1.0, s.sp_Ad)
        s.sp_x.axpby_(1.0, s.x, 1.0, s.sp_d)  # :190
        s.sp_Ax.axpby_(1.0, s.Ax, R(1), s.sp_Ad)  # :199 from m-vectors (tau = 1)
        sp_f_Ax, _ = value_and_gradient_into(self.f, s.sp_Ax, s.sp_grad_f_Ax)  # :200-201
        try:
            sc, _ = self.A.fused_tn_pair(s.sp_grad_f_Ax, s.sp_x, s.grad_f_Az, s.z, s.gamma, self.g,
                                         (s.sp_At_grad_f_Ax, s.sp_y, s.sp_z, s.sp_res, s.sp_Ares),
                                         (s.At_grad_f_Az, s.sp_junk_n[0], s.sp_junk_n[1], s.sp_junk_n[2], s.sp_junk_m), image_of_res=True)
        except ProxGradError as e:
            if e.code != _lib.PG_ERR_UNSUPPORTED:
                raise
            self.speculate = False  # no pair sweep at this column length
            return False
        self.counters["A_passes"] += 1
        self.counters["spec_issued"] = self.counters.get("spec_issued", 0) + 1  # first passes taken ahead (make the one-read claim checkable)
        s.sp_Az.axpby_(1.0, s.sp_Ax, -1.0, s.sp_Ares)  # :210 as A x - A res
        s.sp = {"f_Ax": sp_f_Ax, "g_z": sc[0], "res_stats": (sc[1], sc[2], sc[3]), "gamma": s.gamma}
        return True

    def _step(self, s):
        R = s.x.dtype.type
        FBE_x = R(self._model(s) + s.g_z)  # :174
        sigma = R(self.beta * (R(0.5) / s.gamma) * (R(1) - self.alpha))  # :176
        tol = R(10) * R(np.finfo(R).eps) * (R(1) + abs(FBE_x))
        threshold = R(FBE_x - sigma * self._res_sq(s) + tol)  # :178
        # :170-171 as reference swaps: x_prev / res_prev take the current buffers, x and res are rewritten in full below
        s.x_prev, s.x = s.x, s.x_prev
        s.res_prev, s.res = s.res, s.res_prev
        s.res_stats = None
        if s.img:
            s.Ax_prev, s.Ax = s.Ax, s.Ax_prev
            s.Ares_prev, s.Ares = s.Ares, s.Ares_prev  # A res_prev: the last sweep's product (state.Ares is rewritten below)
        tau_backtracks = 0
        can_update_direction = True
        use_img = False
        spec = getattr(s, "sp", None)
        s.sp = None
        if spec is not None and spec["gamma"] != s.gamma:
            spec = None
            self.counters["spec_discarded"] = self.counters.get("spec_discarded", 0) + 1  # (gamma moved since: the pass is redone)
        while True:  # :183-234
            taken = False
            if can_update_direction and spec is not None:
                # the first pass was taken ahead by the previous step (_speculate): its buffers become the state's
                for name in self.SPEC:
                    a_, b_ = getattr(s, name), getattr(s, "sp_" + name)
                    setattr(s, name, b_), setattr(s, "sp_" + name, a_)
                s.tau, tau_backtracks, use_img = R(1), 0, True
                s.img_steps += 1
                s.f_Ax, s.g_z, s.res_stats = spec["f_Ax"], spec["g_z"], spec["res_stats"]
                self.counters["spec_taken"] = self.counters.get("spec_taken", 0) + 1
                spec = None
                taken = fused = True
            elif can_update_direction:
                use_img = s.img and s.H.images_ready()
                if s.H is not None:  # set_next_direction! :130-138
                    s.H.mul_(s.d, s.res_prev)
                    s.d.axpby_(-1.0, s.d)
                else:
                    s.d.axpby_(-1.0, s.res_prev)
                if use_img:  # A d without reading A: d = -(H res_prev)
                    s.H.images_mul_(s.Ad, s.Ares_prev)
                    s.Ad.axpby_(-1.0, s.Ad)
                s.tau = R(1)  # :189
                s.x.axpby_(1.0, s.x_prev, 1.0, s.d)  # :190
                tau_backtracks = 0
            else:  # :193-196   x = (1 - tau) (x_prev - res_prev) + tau (x_prev + d) = x_prev - (1 - tau) res_prev + tau d
                s.x.axpby_(1.0, s.x_prev, -(R(1) - s.tau), s.res_prev)
                s.x.axpby_(1.0, s.x, s.tau, s.d)
                tau_backtracks += 1
            spec = None  # (only the very first pass can be the one taken ahead)
            if not taken:
                s.img_steps += 1
                if use_img and not (self.refresh_every > 0 and s.img_steps % self.refresh_every == 0):
                    # :199 from m-vectors: A x = A x_prev - (1 - tau) A res_prev + tau A d
                    s.Ax.axpby_(1.0, s.Ax_prev, s.tau, s.Ad)
                    if s.tau != R(1):
                        s.Ax.axpby_(1.0, s.Ax, -(R(1) - s.tau), s.Ares_prev)
                else:
                    self._mul(s.Ax, s.x)  # :199
                s.f_Ax, _ = value_and_gradient_into(self.f, s.Ax, s.grad_f_Ax)  # :200-201
                fused = False
                if self._fused_tn:  # :202-206 and :210 in one read of A (pg_mat_fused_tn)
                    try:
                        sc = self.A.fused_tn(s.grad_f_Ax, s.x, s.gamma, self.g, s.At_grad_f_Ax, s.y, s.z, s.res,
                                             s.Ares if s.img else s.Az, image_of_res=s.img)
                        if s.img:  # :210 as A x - A res, A res being the sweep's product of the residual itself
                            s.Az.axpby_(1.0, s.Ax, -1.0, s.Ares)
                        s.g_z = sc[0]
                        s.res_stats = (sc[1], sc[2], sc[3])  # the sweep's own reductions of this (At_grad, res) pair
                        fused = True
                        self.counters["A_passes"] += 1
                    except ProxGradError as e:
                        if e.code != _lib.PG_ERR_UNSUPPORTED:
                            raise
                        self._fused_tn = False
                if not fused:
                    self._mul_adj(s.At_grad_f_Ax, s.grad_f_Ax)  # :202
                    s.y.axpby_(1.0, s.x, -s.gamma, s.At_grad_f_Ax)  # :204
                    s.g_z = prox_(s.z, self.g, s.y, s.gamma)  # :205
                    s.res.axpby_(1.0, s.x, -1.0, s.z)  # :206
                    s.res_stats = None
            f_Az_upp = self._model(s)  # :208
            if not fused:
                self._mul(s.Az, s.z)  # :210
                if s.img:
                    s.Ares.axpby_(1.0, s.Ax, -1.0, s.Az)
            f_Az, _ = value_and_gradient_into(self.f, s.Az, s.grad_f_Az)  # :211-212
            if self.gamma is None or self.adaptive:  # :213-224
                tol2 = R(10) * R(np.finfo(R).eps) * (R(1) + abs(f_Az))
                if f_Az > f_Az_upp + tol2 and s.gamma >= self.minimum_gamma:
                    s.gamma = R(s.gamma * R(0.5))
                    if s.gamma < self.minimum_gamma:
                        warnings.warn(f"stepsize `gamma` became too small ({s.gamma})")
                    can_update_direction = True
                    if s.H is not None:
                        s.H.reset_()
                    continue
            # (:225, `mul!(state.At_grad_f_Az, adjoint(iter.A), state.grad_f_Az)`, follows the loop: only the last pass's survives)
            FBE_x_new = R(f_Az_upp + s.g_z)  # :227
            if FBE_x_new <= threshold or tau_backtracks >= self.max_backtracks:
                break
            s.tau = R(0) if tau_backtracks >= self.max_backtracks - 1 else R(s.tau / R(2))  # :231
            can_update_direction = False
        if s.H is not None:  # :237 (update_direction_state! :140-148)
            s.x_prev.axpby_(1.0, s.x, -1.0, s.x_prev)
            s.res_prev.axpby_(1.0, s.res, -1.0, s.res_prev)
            s.H.update_(s.x_prev, s.res_prev)
            if s.img:  # A s = tau A d - (1 - tau) A res_prev (:193-195), A y = A res - A res_prev
                if use_img:
                    s.As.axpby_(s.tau, s.Ad, -(R(1) - s.tau), s.Ares_prev)
                else:
                    s.As.axpby_(1.0, s.Ax, -1.0, s.Ax_prev)
                s.Ay.axpby_(1.0, s.Ares, -1.0, s.Ares_prev)
                s.H.images_update_(s.As, s.Ay)
        # :225 -- together with the next iteration's first pass where that can be taken ahead, else as the product it is
        if not self._speculate(s):
            self._mul_adj(s.At_grad_f_Az, s.grad_f_Az)
        return s


def default_stopping_criterion(tol, iteration, state):
    """panocplus.jl:243-244: norm(res / gamma - At_grad_f_Ax + At_grad_f_Az, Inf) <= tol"""
    R = state.res.dtype.type
    t = state.res.similar().axpby_(1.0 / float(state.gamma), state.res, -1.0, state.At_grad_f_Ax)
    t.axpby_(1.0, t, 1.0, state.At_grad_f_Az)
    return R(t.norm_inf()) <= R(tol)


def default_solution(iteration, state):
    """panocplus.jl:245"""
    return state.z


def default_display(it, iteration, state):
    """panocplus.jl:246-253"""
    print("%5d | %.3e | %.3e | %.3e" % (it, state.gamma, state.res.norm_inf() / state.gamma, state.tau))


def PANOCplus(*, maxit=1_000, tol=1e-8, stop=None, solution=default_solution, verbose=False, freq=10,
              display=default_display, **kwargs):
    """panocplus.jl:282-300"""
    if stop is None:
        stop = lambda iteration, state: default_stopping_criterion(tol, iteration, state)
    return IterativeAlgorithm(PANOCplusIteration, maxit=maxit, stop=stop, solution=solution, verbose=verbose,
                              freq=freq, display=display, **kwargs)
