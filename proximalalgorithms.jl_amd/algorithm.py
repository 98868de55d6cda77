"""IterativeAlgorithm -- mirror of src/ProximalAlgorithms.jl:58-123."""
from . import _lib
from ._lib import ProxGradError
from .device import HIPVector


def _like_x0(sol, host_x0):
    """eltype / array kind of the answer follows x0 (test_lasso_small.jl:51); tuples (primal-dual pairs) elementwise"""
    if isinstance(sol, tuple):
        return tuple(_like_x0(v, host_x0) for v in sol)
    return sol.numpy() if (host_x0 and isinstance(sol, HIPVector)) else sol


def graph_iterate(it):
    """Iterate a graph-safe iteration (``init_state`` / ``body``; constant coefficients, no buffer swaps): two plain
    iterations (they allocate every workspace), then the body is RECORDED into a hipGraph and each further iteration
    is one graph launch.  Scalar fields of the state that the body reads back (prox values) are not refreshed in graph
    mode.  Falls back to plain stepping when the context cannot capture (default stream, collective attached) or the
    body turns out to allocate."""
    from ._lib import ProxGradError

    s = it.init_state()
    ctx = it.x0.ctx
    for _ in range(2):
        it.body(s)
        yield s
    graph = None
    try:
        ctx.capture_begin()
        try:
            it.body(s)
        except BaseException:
            ctx.capture_end(abort=True)
            raise
        graph = ctx.capture_end()
    except ProxGradError:
        graph = None
    it.graph = graph  # introspection: None = fell back
    while True:
        if graph is not None:
            graph.launch()
        else:
            it.body(s)
        yield s


class IterativeAlgorithm:
    """Wrapper for an iterator type adding termination and verbosity options
    (src/ProximalAlgorithms.jl:58-112).  Calling it merges the keyword arguments, builds the iterator
    and loops: ``for (k, state) in enumerate(iter)`` -> returns ``(solution, k)`` when
    ``k >= maxit or stop(iter, state)`` (:114-123)."""

    def __init__(self, iterator_type, *, maxit, stop, solution, verbose, freq, display, device_loop=None, graph=False,
                 **kwargs):
        # device_loop = (tol, check_every): run the loop with the DEFAULT stopping rule inside the library instead of
        # stepping from the host (fused engines only): one launch for launch-bound sizes (pg_iter_run_small /
        # pg_iter_run_coop), else the in-library loop (pg_iter_run / pg_iter_run_batched)
        self.device_loop = device_loop
        self.graph = bool(graph)  # record the iteration body into a hipGraph (graph-safe iterations only)
        self.iterator_type = iterator_type
        self.maxit = int(maxit)
        self.stop = stop
        self.solution = solution
        self.verbose = bool(verbose)
        self.freq = int(freq)
        self.display = display
        self.kwargs = kwargs

    def __call__(self, **kwargs):
        merged = dict(self.kwargs)
        merged.update(kwargs)
        x0 = merged.get("x0")
        host_x0 = x0 is not None and not isinstance(x0, HIPVector)
        it = self.iterator_type(**merged)
        if (self.device_loop is not None and getattr(it, "engine", None) == "fused" and not self.verbose
                and hasattr(it, "device_run")):
            state, k = it.device_run(self.maxit, *self.device_loop)  # iterations that carry their own in-library loop
            return _like_x0(self.solution(it, state), host_x0), k
        if self.device_loop is not None and getattr(it, "engine", None) == "fused" and not self.verbose:
            tol, check_every = self.device_loop
            gen = iter(it)
            state = next(gen)  # Base.iterate(iter): k = 1
            fused = it._fused
            A = it.f.A
            adaptive = bool(it.adaptive)
            nbytes = A.m * A.n * A.dtype.itemsize
            # (the one-launch solvers take scalar IndBox bounds only)
            unsharded = A.m > 0 and A.n > 0 and it.f.comm is None and fused._g_vectors is None
            # measured crossovers (tests/tools/bench_small.py, profiles/): one workgroup up to ~8k elements; the cooperative
            # multi-workgroup kernel while A is a few MiB (its barriers beat launches + host round trips: up to ~10 MiB
            # with the adaptive step, ~8 MiB against per-iteration syncs, ~3 MiB against the batched fixed-step loop);
            # beyond that the streaming kernels driven from the host
            coop_rows = 3 * (-(-A.m // 64) * 64) * A.dtype.itemsize <= 96 * 1024
            if unsharded and A.m * A.n <= 8192:
                k, _ = fused.run_small(1, self.maxit, tol)
            elif unsharded and coop_rows and nbytes <= ((10 << 20) if adaptive else ((3 << 20) if check_every > 1 else (8 << 20))):
                k, _ = fused.run_coop(1, self.maxit, tol)
            elif check_every > 1 and not adaptive:
                try:
                    k, _ = fused.run(1, self.maxit, tol, check_every=check_every)
                except ProxGradError as e:
                    # A team sweep lost inside a batch (PG_ERR_TIMEOUT, seen at the batch's one read-back) cannot be redone:
                    # the iterations behind it are already enqueued.  Start over from x0 with the per-iteration loop, which
                    # redoes a lost sweep with two sweeps and carries on (pg_iter_run; csrc/pg_iter.hip).
                    # (PG_ERR_UNSUPPORTED at that read-back: with column shards some rank's sweep was refused at launch; every
                    # rank sees it in the same batch and has left the single-sweep mode.)
                    if e.code not in (_lib.PG_ERR_TIMEOUT, _lib.PG_ERR_UNSUPPORTED):
                        raise
                    import warnings

                    warnings.warn("a long-column sweep %s inside a batch of %d iterations; the solve restarts with one "
                                  "synchronisation per iteration"
                                  % ("timed out" if e.code == _lib.PG_ERR_TIMEOUT else "was refused", check_every))
                    it.counters["sweep_fallbacks"] = it.counters.get("sweep_fallbacks", 0) + 1
                    fused.init(it.x0)  # (the iteration holds x0 as a device vector and never writes it)
                    k, _ = fused.run(1, self.maxit, tol)
            else:
                k, _ = fused.run(1, self.maxit, tol)
            state._invalidate()
            return _like_x0(self.solution(it, state), host_x0), k
        steps = graph_iterate(it) if (self.graph and getattr(it, "graph_safe", False)) else it
        for k, state in enumerate(steps, start=1):
            if k >= self.maxit or self.stop(it, state):
                if self.verbose:
                    self.display(k, it, state)
                return _like_x0(self.solution(it, state), host_x0), k
            if self.verbose and k % self.freq == 0:
                self.display(k, it, state)
