"""Line-search helpers on device vectors -- mirror of src/utilities/fb_tools.jl (generic-operator path).
The fused path implements the same logic inside libproxgrad_hip (csrc/pg_iter.hip)."""
import warnings

import numpy as np

from .operators import prox_, value_and_gradient


def f_model(f_x, grad_f_x, res, L):
    """fb_tools.jl:3-5"""
    R = res.dtype.type
    return R(R(f_x) - grad_f_x.dot(res) + (R(L) / R(2)) * res.norm() ** 2)


def lower_bound_smoothness_constant(f, x, grad_f_x=None):
    """fb_tools.jl:7-19 with A = I"""
    R = x.dtype.type
    if grad_f_x is None:
        _, grad_f_x = value_and_gradient(f, x)
    xeps = x.similar().add_scalar_(x, 1.0)
    _, grad_eps = value_and_gradient(f, xeps)
    diff = x.similar().axpby_(1.0, grad_eps, -1.0, grad_f_x)
    return R(diff.norm() / R(np.sqrt(x.n)))


def backtrack_stepsize_(gamma, f, g, x, f_x, grad_f_x, y, z, g_z, res, grad_f_z=None, *, alpha=1.0,
                        minimum_gamma=1e-7, reduce_gamma=0.5, counters=None):
    """backtrack_stepsize!  fb_tools.jl:24-63 with A === nothing.  Returns (gamma, g_z, f_z, f_z_upp)."""
    R = x.dtype.type
    gamma, alpha, minimum_gamma, reduce_gamma = R(gamma), R(alpha), R(minimum_gamma), R(reduce_gamma)
    eps = R(np.finfo(R).eps)
    f_z_upp = f_model(f_x, grad_f_x, res, alpha / gamma)
    f_z, grad_tmp = value_and_gradient(f, z)
    tol = R(10) * eps * (R(1) + abs(f_z))
    nbt = 0
    while f_z > f_z_upp + tol and gamma >= minimum_gamma:
        gamma = R(gamma * reduce_gamma)
        y.axpby_(1.0, x, -gamma, grad_f_x)
        g_z = prox_(z, g, y, gamma)
        res.axpby_(1.0, x, -1.0, z)
        f_z_upp = f_model(f_x, grad_f_x, res, alpha / gamma)
        f_z, grad_tmp = value_and_gradient(f, z)
        tol = R(10) * eps * (R(1) + abs(f_z))
        nbt += 1
    if grad_f_z is not None:
        grad_f_z.copy_from(grad_tmp)
    if gamma < minimum_gamma:
        warnings.warn(f"stepsize `gamma` became too small ({gamma})")
    if counters is not None:
        counters["backtracks"] = counters.get("backtracks", 0) + nbt
    return gamma, g_z, f_z, f_z_upp
