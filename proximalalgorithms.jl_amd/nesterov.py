"""Extrapolation-coefficient sequences (host scalars in the working precision R).
Mirror of src/accel/nesterov.jl."""
import itertools

import numpy as np


class FixedNesterovSequence:
    """nesterov.jl:1-20: t0 = 1 ; t+ = (1 + sqrt(1 + 4 t^2)) / 2 ; yields (t - 1) / t+."""

    def __init__(self, R=np.float64):
        self.R = np.dtype(R).type

    def __iter__(self):
        R = self.R
        t = R(1)
        while True:
            t_next = R((R(1) + np.sqrt(R(1) + R(4) * t * t)) / R(2))
            yield R((t - R(1)) / t_next)
            t = t_next


class SimpleNesterovSequence:
    """nesterov.jl:22-39: (k - 1) / (k + 2), k >= 1."""

    def __init__(self, R=np.float64):
        self.R = np.dtype(R).type

    def __iter__(self):
        R = self.R
        k = 1
        while True:
            yield R(R(k - 1) / R(k + 2))
            k += 1


class ConstantNesterovSequence:
    """nesterov.jl:51-54: repeated((1 - sqrt(m s)) / (1 + sqrt(m s)))."""

    def __init__(self, m, stepsize):
        R = np.asarray(m).dtype.type if isinstance(m, np.generic) else np.float64
        self.R = R
        self.m, self.stepsize = R(m), R(stepsize)

    def __iter__(self):
        R = self.R
        k_inverse = R(self.m * self.stepsize)
        return itertools.repeat(R((R(1) - np.sqrt(k_inverse)) / (R(1) + np.sqrt(k_inverse))))


class AdaptiveNesterovSequence:
    """nesterov.jl:56-80 (state) and next! :89-103."""

    def __init__(self, m, R=None):
        R = np.dtype(R).type if R is not None else (type(m) if isinstance(m, np.generic) else np.float64)
        self.R = R
        self.m = R(m)
        self.stepsize = -R(1)
        self.theta = -R(1)

    def next(self, stepsize):
        R = self.R
        stepsize = R(stepsize)
        if self.stepsize < 0:
            self.stepsize = stepsize
            self.theta = R(np.sqrt(self.m * stepsize)) if self.m > 0 else R(1)
        b = R(self.theta**2 / self.stepsize - self.m)
        delta = R(b**2 + R(4) * (self.theta**2) / (self.stepsize * stepsize))
        theta = R(stepsize * (-b + np.sqrt(delta)) / R(2))
        beta = R(stepsize * self.theta * (R(1) - self.theta) / (self.stepsize * theta + stepsize * self.theta**2))
        self.stepsize = stepsize
        self.theta = theta
        return beta


def next_(seq, stepsize):
    """ProximalAlgorithms.next!(seq, stepsize)"""
    return seq.next(stepsize)


class NesterovExtrapolation:
    """nesterov.jl:105-113: acceleration-style tag; ``initialize(x)`` is a stateful iterator over the sequence type
    (default SimpleNesterovSequence) in the precision of x."""

    def __init__(self, sequence_type=SimpleNesterovSequence):
        self.sequence_type = sequence_type

    def initialize(self, x):
        return iter(self.sequence_type(x.dtype.type))
