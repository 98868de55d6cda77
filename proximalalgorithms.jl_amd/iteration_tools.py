"""IterationTools -- mirror of src/utilities/iteration_tools.jl (host-side iterator combinators: halt, tee, sample,
stopwatch, loop).  They wrap any iterable, in particular the iteration objects of this package, whose states live on the
device; nothing here touches device memory."""
import time


class _Wrapped:
    def __init__(self, it):
        self.iter = it

    def __len__(self):
        return len(self.iter)


class HaltingIterable(_Wrapped):
    """iteration_tools.jl:7-37: yields the elements of ``iter`` up to and INCLUDING the first for which fun is true"""

    def __init__(self, it, fun):
        super().__init__(it)
        self.fun = fun

    def __iter__(self):
        for x in self.iter:
            yield x
            if self.fun(x):
                return


class TeeIterable(_Wrapped):
    """iteration_tools.jl:41-62: calls fun on every element (side effects), passes it on"""

    def __init__(self, it, fun):
        super().__init__(it)
        self.fun = fun

    def __iter__(self):
        for x in self.iter:
            self.fun(x)
            yield x


class SamplingIterable(_Wrapped):
    """iteration_tools.jl:66-99: every period-th element, and the last one if the length is not a multiple"""

    def __init__(self, it, period):
        super().__init__(it)
        self.period = int(period)
        if self.period < 1:
            raise ValueError("period must be positive")

    def __len__(self):
        q, r = divmod(len(self.iter), self.period)
        return q if r == 0 else q + 1

    def __iter__(self):
        last, k = None, 0
        for x in self.iter:
            last, k = x, k + 1
            if k == self.period:
                yield x
                k = 0
        if k:
            yield last


class StopwatchIterable(_Wrapped):
    """iteration_tools.jl:103-128: pairs (nanoseconds since the first iterate call, element)"""

    def __iter__(self):
        t0 = time.perf_counter_ns()
        for x in self.iter:
            yield time.perf_counter_ns() - t0, x


def halt(it, fun):
    return HaltingIterable(it, fun)


def tee(it, fun):
    return TeeIterable(it, fun)


def sample(it, period):
    return SamplingIterable(it, period)


def stopwatch(it):
    return StopwatchIterable(it)


def loop(it):
    """iteration_tools.jl:132-140: run to exhaustion, return the last element (error on an empty iterable, like the
    reference's destructuring of `nothing`)"""
    it = iter(it)
    try:
        output = next(it)
    except StopIteration:
        raise TypeError("loop needs a non-empty iterable") from None
    for output in it:
        pass
    return output
