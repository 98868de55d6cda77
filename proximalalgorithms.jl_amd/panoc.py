"""PANOC -- mirror of src/algorithms/panoc.jl (BASELINE config 4): minimize f(A x) + g(x).

The iteration body is the reference's, statement by statement, on device vectors; every array statement is a HIP
kernel of libproxgrad_hip (the two GEMV orientations for `mul!` with A and A', the loss kernels, prox, AXPYs,
reductions, and the device L-BFGS two-loop recursion).  At config-4 size (16384 x 10^6) an accepted iteration is ONE
pass over a 61 GiB matrix (the single sweep; the `mul!(state.Ad, iter.A, state.d)` of panoc.jl:180 comes out of the
L-BFGS operator's image slab without reading A), so host-side sequencing of the ~45 small kernels is noise.
"""
import warnings

import numpy as np

from . import _lib
from ._lib import ProxGradError
from .algorithm import IterativeAlgorithm
from .device import HIPMatrix, HIPVector, as_hipvector
from .lbfgs import LBFGS, LBFGSOperator
from .operators import Zero, prox_, value_and_gradient


class NoAcceleration:
    """src/accel/noaccel.jl:1-5"""

    def initialize(self, x):
        return None


class _Identity:
    """LinearAlgebra.I as the default `A` (panoc.jl:41)."""

    def mul(self, x, out=None):
        if out is None:
            return x.copy()
        return out.copy_from(x)

    mul_adjoint = mul


def _f_model(f_x, grad, res, L):
    """fb_tools.jl:3-5"""
    R = res.dtype.type
    return R(R(f_x) - grad.dot(res) + (R(L) / R(2)) * res.norm() ** 2)


class PANOCState:
    """panoc.jl:56-82"""

    pass


class PANOCIteration:
    """panoc.jl:39-52 (keyword constructor), Base.iterate :87-112 / :138-255."""

    def __init__(self, *, f=None, A=None, g=None, x0, alpha=0.95, beta=0.5, Lf=None, gamma=None, adaptive=None,
                 minimum_gamma=1e-7, max_backtracks=20, directions=None, single_sweep=True, images=True,
                 refresh_every=0, pair_trials=True, trio_trials=True, speculate=True, gamma_candidates=3):
        self.f = f if f is not None else Zero()
        if A is None:
            A = _Identity()
        elif not isinstance(A, (HIPMatrix, _Identity)):
            A = HIPMatrix.from_numpy(A)
        self.A = A
        self.g = g if g is not None else Zero()
        self.x0 = as_hipvector(x0, getattr(A, "ctx", None))
        R = self.x0.dtype.type
        self.alpha, self.beta = R(alpha), R(beta)
        self.Lf = Lf
        self.gamma = gamma if gamma is not None else (None if Lf is None else self.alpha / R(Lf))  # :47
        self.adaptive = (self.gamma is None) if adaptive is None else bool(adaptive)  # :48
        self.minimum_gamma = R(minimum_gamma)
        self.max_backtracks = int(max_backtracks)
        self.directions = directions if directions is not None else LBFGS(5)  # :51
        self.counters = {"A_passes": 0}
        # ZeroFPR: further trial points of the line search in the sweep of A that evaluates one (zerofpr.py): pair_trials: the next one
        # (pg_mat_fused_tn_pair); trio_trials: the next two (pg_mat_fused_tn_trio, the default where the kernel applies).  Both False:
        # one trial point per sweep, the reference's count.
        self.pair_trials = bool(pair_trials)
        self.trio_trials = bool(trio_trials)
        # PANOCplus: its second pass over A (panocplus.jl:225) rides in the next iteration's first sweep, taken ahead (panocplus.py)
        self.speculate = bool(speculate)
        # the step-size search (_backtrack_stepsize): up to this many candidates gamma r, gamma r^2, ... per read of A (1: the reference's
        # one product per candidate)
        self._gamma_multi = int(gamma_candidates) >= 3 and isinstance(A, HIPMatrix) and hasattr(A, "mul_multi")
        # A' grad f(A x) (:184), the forward-backward step (:197-199) and the A z of the next line search (fb_tools.jl:43)
        # in ONE read of A (pg_mat_fused_tn) when A is a device matrix and g one of the fused prox kinds
        self._fused_tn = bool(single_sweep) and isinstance(A, HIPMatrix) and hasattr(self.g, "g_kind") and \
            not (hasattr(self.g, "_scalar") and not self.g._scalar)
        # `mul!(state.Ad, iter.A, state.d)` (:180) WITHOUT reading A when the directions are L-BFGS: d = -H res, and
        # A res = A x - A z are m-vectors the iteration holds (the sweep leaves A z), so A d follows from the images A s_i,
        # A y_i kept next to the stored pairs (pg_lbfgs_images_*); the images of a new pair are again differences of
        # m-vectors.  What the recurrence costs: state.Ax (already a running sum in the reference, :181 / :187) now adds
        # image-derived A d instead of products -- Float32 drift measured in tests/test_gpu_parity.py
        # (test_panoc_image_recurrence_drift); refresh_every = K > 0 replaces every K-th update of A x by a product.
        self._images = bool(images) and isinstance(A, HIPMatrix)
        self.refresh_every = int(refresh_every)

    # mul! with A / A' (counted: each is one full read of A)
    def _mul(self, out, x):
        self.counters["A_passes"] += 1
        return self.A.mul(x, out)

    def _mul_adj(self, out, r):
        self.counters["A_passes"] += 1
        return self.A.mul_adjoint(r, out)

    def _mul_start(self, x):
        """`mul!(state.Ax, iter.A, state.x)` at the start (panoc.jl:89, zerofpr.jl:87, panocplus.jl:89): A 0 = 0 exactly, so a start
        from zero -- the usual one -- does not read A for it (one n-vector reduction to see that it is zero)."""
        if isinstance(self.A, HIPMatrix) and float(x.norm_inf()) == 0.0:
            return HIPVector.zeros(self.A.m, x.dtype, x.ctx)
        return self._mul(None, x)

    def _model(self, s, gamma=None):  # :84-85
        """f_model at the current (At_grad_f_Ax, res) pair.  When the pair comes out of the single sweep, the sweep's own
        reductions <At_grad, res> and ||res||^2 are reused (no further kernels, no host round trips)."""
        gamma = s.gamma if gamma is None else gamma
        st = getattr(s, "res_stats", None)
        if st is not None:
            R = s.x.dtype.type
            return R(R(s.f_Ax) - st[1] + (R(self.alpha / gamma) / R(2)) * st[2])
        return _f_model(s.f_Ax, s.At_grad_f_Ax, s.res, self.alpha / gamma)

    def _res_sq(self, s):
        st = getattr(s, "res_stats", None)
        return st[2] if st is not None else s.res.norm() ** 2

    def _lower_bound_smoothness_constant(self, x, grad_f_Ax):
        """fb_tools.jl:7-12"""
        R = x.dtype.type
        xeps = x.similar().add_scalar_(x, 1.0)
        Axeps = self._mul(None, xeps)
        _, grad_eps = value_and_gradient(self.f, Axeps)
        diff = grad_eps.axpby_(1.0, grad_eps, -1.0, grad_f_Ax)
        return R(self._mul_adj(None, diff).norm() / R(np.sqrt(x.n)))

    def _take_Az(self, s, z, Az):
        """`mul!(Az, A, z)` (fb_tools.jl:43, panoc.jl:210, zerofpr.jl:167) for the forward-backward point z: out of the last
        sweep when that was taken at z -- it left A z itself, or A res = A (x - z), and then A z = A x - A res with the
        very A x that f(A x) was evaluated at -- else a product.  Returns whether the sweep served."""
        if getattr(s, "Az_next_valid", False) and getattr(s, "Az_next_of", None) is z:
            if getattr(s, "Az_next_is_res", False):
                Az.axpby_(1.0, s.Ax, -1.0, s.Az_next)
            else:
                Az.copy_from(s.Az_next)
            return True
        self._mul(Az, z)
        return False

    def _gamma_buffers(self, s, k, z, Az):
        """(y, z, res, Az) of the k-th candidate evaluated ahead by the step-size search (allocated on first use)"""
        store = s.__dict__.setdefault("_gamma_ahead", {})
        if k not in store:
            store[k] = (s.y.similar(), z.similar(), s.res.similar(), Az.similar())
        return store[k]

    def _backtrack_stepsize(self, s, z, g_z, Az, grad_f_Az):
        """backtrack_stepsize!  fb_tools.jl:24-63 with the linear map A and alpha = iter.alpha; z / Az / grad_f_Az
        are the forward-backward point and its images (state.z | state.xbar ...).  Returns (gamma, g_z, f_Az, f_Az_upp)."""
        R = s.x.dtype.type
        eps = R(np.finfo(R).eps)
        gamma, reduce_gamma = R(s.gamma), R(0.5)
        f_Az_upp = self._model(s, gamma)  # :42
        self._take_Az(s, z, Az)  # :43
        f_Az, _ = value_and_gradient_into(self.f, Az, grad_f_Az)  # :44 (grad kept: :56-58)
        tol = R(10) * eps * (R(1) + abs(f_Az))
        # Every halving of gamma costs the reference one product `mul!(Az, A, z)` (:52) -- a read of A -- and its candidates gamma / 2,
        # gamma / 4, gamma / 8 differ in the n-vector z alone.  So the first halving of a search forms the next two candidates as well
        # (two more forward-backward steps on n-vectors) and takes the three images in ONE read of A (pg_mat_mul_multi: per vector the
        # same multiply-adds in the same order as the single product, bit-identical images); a candidate evaluated ahead is looked at
        # only after the one before it was rejected, exactly where the reference would have formed it: the same decisions, bit for bit.
        ahead = []  # [(gamma_k, g(z_k), (y_k, z_k, res_k, Az_k))]: candidates behind the current one whose images are already there
        while f_Az > f_Az_upp + tol and gamma >= self.minimum_gamma:  # :46
            s.Az_next_valid = False
            gamma = R(gamma * reduce_gamma)
            if ahead and ahead[0][0] == gamma:  # evaluated ahead: its vectors become the state's, no read of A
                _, g_z, (y_k, z_k, res_k, Az_k) = ahead.pop(0)
                s.y.copy_from(y_k), z.copy_from(z_k), s.res.copy_from(res_k), Az.copy_from(Az_k)
                self.counters["gamma_candidates_taken"] = self.counters.get("gamma_candidates_taken", 0) + 1
            else:
                ahead = []
                s.y.axpby_(1.0, s.x, -gamma, s.At_grad_f_Ax)
                g_z = prox_(z, self.g, s.y, gamma)
                s.res.axpby_(1.0, s.x, -1.0, z)
                g_k = gamma
                for k in range(2 if self._gamma_multi else 0):
                    if not g_k >= self.minimum_gamma:  # (the loop would not get to a further candidate)
                        break
                    g_k = R(g_k * reduce_gamma)
                    bufs = self._gamma_buffers(s, k, z, Az)
                    bufs[0].axpby_(1.0, s.x, -g_k, s.At_grad_f_Ax)
                    gz_k = prox_(bufs[1], self.g, bufs[0], g_k)
                    bufs[2].axpby_(1.0, s.x, -1.0, bufs[1])
                    ahead.append((g_k, gz_k, bufs))
                if ahead:
                    try:
                        self.A.mul_multi([z] + [b_[1] for _, _, b_ in ahead], [Az] + [b_[3] for _, _, b_ in ahead])
                        self.counters["A_passes"] += 1
                        self.counters["gamma_candidates_ahead"] = self.counters.get("gamma_candidates_ahead", 0) + len(ahead)
                    except ProxGradError as e:
                        if e.code != _lib.PG_ERR_UNSUPPORTED:
                            raise
                        self._gamma_multi, ahead = False, []  # single products only for this operator
                        self._mul(Az, z)
                else:
                    self._mul(Az, z)
            s.res_stats = s.res_inf = None
            f_Az_upp = _f_model(s.f_Ax, s.At_grad_f_Ax, s.res, self.alpha / gamma)
            f_Az, _ = value_and_gradient_into(self.f, Az, grad_f_Az)
            tol = R(10) * eps * (R(1) + abs(f_Az))
        if gamma < self.minimum_gamma:
            warnings.warn(f"stepsize `gamma` became too small ({gamma})")
        return gamma, g_z, f_Az, f_Az_upp

    def _init(self):
        R = self.x0.dtype.type
        s = PANOCState()
        s.x = self.x0.copy()  # :88
        s.Ax = self._mul_start(s.x)  # :89
        s.grad_f_Ax = s.Ax.similar()
        s.f_Ax, _ = value_and_gradient_into(self.f, s.Ax, s.grad_f_Ax)  # :90
        if self.gamma is None:  # :91-94
            s.gamma = R(self.alpha / self._lower_bound_smoothness_constant(s.x, s.grad_f_Ax))
        else:
            s.gamma = R(self.gamma)
        s.At_grad_f_Ax = self._mul_adj(None, s.grad_f_Ax)  # :95
        s.y = s.x.similar().axpby_(1.0, s.x, -s.gamma, s.At_grad_f_Ax)  # :96
        s.z = s.x.similar()
        s.g_z = prox_(s.z, self.g, s.y, s.gamma)  # :97
        s.res = s.x.similar().axpby_(1.0, s.x, -1.0, s.z)
        s.H = self.directions.initialize(s.x)  # :109
        s.tau = R(0)
        for name in ("x_prev", "res_prev", "d", "x_d", "At_grad_f_Ax_d", "z_curr", "At_grad_f_Az"):
            setattr(s, name, s.x.similar())
        for name in ("Ad", "Ax_d", "grad_f_Ax_d", "Az", "grad_f_Az", "Az_next"):
            setattr(s, name, s.Ax.similar())
        s.Az_next_valid, s.Az_next_of, s.Az_next_is_res = False, None, False
        s.f_Ax_d = R(0)
        s.res_inf = None
        self._images_init(s)
        return s

    def _images_init(self, s):
        """the image slab of the L-BFGS operator and the m-vectors the image recurrences need"""
        s.img = self._images and isinstance(s.H, LBFGSOperator)
        s.img_steps = 0
        if s.img:
            s.H.images_enable(s.Ax.n)
            for name in ("Ares", "As", "Ay"):
                setattr(s, name, s.Ax.similar())

    def _step(self, s):
        R = s.x.dtype.type
        inf = R(np.inf)
        f_Az, a, b, c = inf, inf, inf, inf  # :139
        have_Az = False  # state.Az holds A z for the current z
        # the last sweep left A res for this very (x, z) pair (the image slab's input, below)
        sweep_Ares = s.img and s.Az_next_valid and s.Az_next_of is s.z and s.Az_next_is_res
        if self.adaptive:  # :141-163
            gamma_prev = s.gamma
            s.gamma, s.g_z, f_Az, f_Az_upp = self._backtrack_stepsize(s, s.z, s.g_z, s.Az, s.grad_f_Az)
            have_Az = True
            if s.gamma != gamma_prev and s.H is not None:
                s.H.reset_()
        else:
            f_Az_upp = self._model(s)  # :165
            if s.Az_next_valid and s.Az_next_of is s.z:
                have_Az = self._take_Az(s, s.z, s.Az)  # the last sweep serves :209-211 (and the images below)
        FBE_x = R(f_Az_upp + s.g_z)  # :169
        use_img = s.img = s.img and self._fused_tn
        if s.img:
            if not have_Az:
                self._mul(s.Az, s.z)  # (first iteration with a fixed step) one read here instead of the A d below
                have_Az = True
            if sweep_Ares and s.Az_next_valid:  # (a rejected gamma recomputes z and invalidates the sweep's output)
                s.Ares.copy_from(s.Az_next)  # A res as a product of the residual itself
            else:
                s.Ares.axpby_(1.0, s.Ax, -1.0, s.Az)  # A res = A x - A z
            use_img = s.H.images_ready()
        if s.H is not None:  # :172 (set_next_direction! :114-117)
            s.H.mul_(s.d, s.res)
            s.d.axpby_(-1.0, s.d)
        else:
            s.d.axpby_(-1.0, s.res)
        sigma = R(self.beta * (R(0.5) / s.gamma) * (R(1) - self.alpha))  # :195
        tol = R(10) * R(np.finfo(R).eps) * (R(1) + abs(FBE_x))  # :196
        threshold = R(FBE_x - sigma * self._res_sq(s) + tol)  # :197 (the residual of the CURRENT point)
        # :175-176, :188-192 -- the reference's copyto! statements are reference swaps here: x_prev / res_prev take the
        # current buffers, the trial point x + d (tau = 1) is formed directly in x / Ax / grad_f_Ax / At_grad_f_Ax, and
        # the separate copies the line search interpolates from (x_d, Ax_d, grad_f_Ax_d, At_grad_f_Ax_d) are only
        # materialised if a backtrack actually happens (_materialize_trial).  Same values in every named vector whenever
        # the reference reads it; seven device copies fewer per accepted step.
        s.x_prev, s.x = s.x, s.x_prev
        s.res_prev, s.res = s.res, s.res_prev
        s.res_stats = s.res_inf = None
        s.tau = R(1)  # :179
        if use_img:  # :180 without reading A: d = -(H res)  =>  A d = -(image of A res under the two-loop coefficients)
            s.H.images_mul_(s.Ad, s.Ares)
            s.Ad.axpby_(-1.0, s.Ad)
        else:
            self._mul(s.Ad, s.d)  # :180
        s.x.axpby_(1.0, s.x_prev, 1.0, s.d)  # :182, :188
        s.img_steps += 1
        if use_img and self.refresh_every > 0 and s.img_steps % self.refresh_every == 0:
            self._mul(s.Ax, s.x)  # the running sum A x restarted from a product
        else:
            s.Ax.axpby_(1.0, s.Ax, 1.0, s.Ad)  # :183, :189
        s.f_Ax, _ = value_and_gradient_into(self.f, s.Ax, s.grad_f_Ax)  # :184-185, :190, :193
        s.f_Ax_d = s.f_Ax
        s.z_curr, s.z = s.z, s.z_curr  # :192 (z is rewritten below)
        fused = False
        if self._fused_tn:
            # :186 and :199-201 in one read of A, which also leaves A z for the next iteration's line search
            try:
                sc = self.A.fused_tn(s.grad_f_Ax, s.x, s.gamma, self.g, s.At_grad_f_Ax, s.y, s.z, s.res, s.Az_next,
                                     image_of_res=s.img)  # with the image slab: A res in place of A z (_take_Az)
                s.g_z = sc[0]
                fused = True
            except ProxGradError as e:
                if e.code != _lib.PG_ERR_UNSUPPORTED:
                    raise  # a HIP / allocation / argument failure is not a reason to change path
                self._fused_tn = False  # shape outside the kernel's range: separate sweeps from now on
        if fused:
            self.counters["A_passes"] += 1
            s.Az_next_valid, s.Az_next_of, s.Az_next_is_res = True, s.z, s.img
            s.res_stats, s.res_inf = (sc[1], sc[2], sc[3]), sc[1]  # norm(res, Inf), <At_grad, res>, ||res||^2 of this pair
        else:
            self._mul_adj(s.At_grad_f_Ax, s.grad_f_Ax)  # :186, :191
            s.y.axpby_(1.0, s.x, -s.gamma, s.At_grad_f_Ax)  # :199
            s.g_z = prox_(s.z, self.g, s.y, s.gamma)  # :200
            s.res.axpby_(1.0, s.x, -1.0, s.z)  # :201
        FBE_x_new = R(self._model(s) + s.g_z)  # :202
        quad = getattr(self.f, "is_generalized_quadratic", False)
        for k in range(1, self.max_backtracks + 1):  # :204-250
            if FBE_x_new <= threshold:
                break
            if k == 1:
                self._materialize_trial(s)
            s.Az_next_valid = False  # z is about to be recomputed
            if np.isinf(f_Az) and not have_Az:  # :209-211
                self._mul(s.Az, s.z_curr)
                have_Az = True
            s.tau = R(0) if k >= self.max_backtracks else R(s.tau / R(2))  # :213
            s.x.axpby_(s.tau, s.x_d, R(1) - s.tau, s.z_curr)  # :214
            s.Ax.axpby_(s.tau, s.Ax_d, R(1) - s.tau, s.Az)  # :215
            if quad:  # :217-237
                if np.isinf(f_Az):
                    f_Az, _ = value_and_gradient_into(self.f, s.Az, s.grad_f_Az)
                if np.isinf(c):
                    self._mul_adj(s.At_grad_f_Az, s.grad_f_Az)
                    c = f_Az
                    b = R(s.Ax_d.dot(s.grad_f_Az) - s.Az.dot(s.grad_f_Az))
                    a = R(s.f_Ax_d - b - c)
                s.f_Ax = R(a * s.tau**2 + b * s.tau + c)
                s.grad_f_Ax.axpby_(s.tau, s.grad_f_Ax_d, R(1) - s.tau, s.grad_f_Az)
                s.At_grad_f_Ax.axpby_(s.tau, s.At_grad_f_Ax_d, R(1) - s.tau, s.At_grad_f_Az)
            else:  # :238-244
                s.f_Ax, _ = value_and_gradient_into(self.f, s.Ax, s.grad_f_Ax)
                self._mul_adj(s.At_grad_f_Ax, s.grad_f_Ax)
            s.y.axpby_(1.0, s.x, -s.gamma, s.At_grad_f_Ax)  # :246
            s.g_z = prox_(s.z, self.g, s.y, s.gamma)  # :247
            s.res.axpby_(1.0, s.x, -1.0, s.z)  # :248
            s.res_stats = s.res_inf = None
            FBE_x_new = R(self._model(s) + s.g_z)  # :249
        if s.H is not None:  # :252 (update_direction_state! :122-126)
            s.x_prev.axpby_(1.0, s.x, -1.0, s.x_prev)
            s.res_prev.axpby_(1.0, s.res, -1.0, s.res_prev)
            s.H.update_(s.x_prev, s.res_prev)
            if s.img:  # the images of the pair, from m-vectors: A s = tau A d + (1 - tau) A (z_curr - x_prev) (:214),
                # A y = A res+ - A res_prev, both residual images being products of the residuals themselves
                if not (s.Az_next_valid and s.Az_next_of is s.z and s.Az_next_is_res):  # z was recomputed by the tau search
                    self._mul(s.Az_next, s.res)  # (with an adaptive step this also serves the next line search, _take_Az)
                    s.Az_next_valid, s.Az_next_of, s.Az_next_is_res = True, s.z, True
                s.As.axpby_(s.tau, s.Ad, -(R(1) - s.tau), s.Ares)
                s.Ay.axpby_(1.0, s.Az_next, -1.0, s.Ares)
                s.H.images_update_(s.As, s.Ay)
        return s

    @staticmethod
    def _materialize_trial(s):
        """the tau = 1 endpoint of the line search as separate vectors (panoc.jl:182-186), made only when a backtrack
        is about to overwrite x / Ax / grad_f_Ax / At_grad_f_Ax"""
        s.x_d.copy_from(s.x)
        s.Ax_d.copy_from(s.Ax)
        s.grad_f_Ax_d.copy_from(s.grad_f_Ax)
        s.At_grad_f_Ax_d.copy_from(s.At_grad_f_Ax)

    def __iter__(self):
        s = self._init()
        yield s
        while True:
            yield self._step(s)


def value_and_gradient_into(f, u, grad_out):
    """value_and_gradient(f, u) with the gradient copied into a preallocated vector (`state.grad .= grad`)."""
    try:
        v, g = f.value_and_gradient(u, out=grad_out)
    except TypeError:
        v, g = f.value_and_gradient(u)
    if g is not grad_out and g.ptr != grad_out.ptr:
        grad_out.copy_from(g)
    return v, grad_out


def default_stopping_criterion(tol, iteration, state):
    """panoc.jl:256-257"""
    R = state.res.dtype.type
    res_inf = state.res_inf if getattr(state, "res_inf", None) is not None else state.res.norm_inf()
    return R(res_inf) / R(state.gamma) <= R(tol)


def default_solution(iteration, state):
    """panoc.jl:258"""
    return state.z


def default_display(it, iteration, state):
    """panoc.jl:259-266"""
    print("%5d | %.3e | %.3e | %.3e" % (it, state.gamma, state.res.norm_inf() / state.gamma, state.tau))


def PANOC(*, maxit=1_000, tol=1e-8, stop=None, solution=default_solution, verbose=False, freq=10,
          display=default_display, **kwargs):
    """panoc.jl:297-315"""
    if stop is None:
        stop = lambda iteration, state: default_stopping_criterion(tol, iteration, state)
    return IterativeAlgorithm(PANOCIteration, maxit=maxit, stop=stop, solution=solution, verbose=verbose, freq=freq,
                              display=display, **kwargs)
