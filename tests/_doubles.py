"""Test doubles shared by the GPU tests (not part of the product package)."""


class ScaleComm:
    """Test double for the collective on a single GPU: emulates ``world_size`` ranks holding IDENTICAL row
    shards (the SUM all-reduce of identical buffers is a multiplication by world_size), using the library's
    own axpby kernel on the payload.  Exercises the C-side pack / reduce / unpack path without RCCL."""

    def __init__(self, world_size, overlap=False):
        self.world_size = int(world_size)
        self.overlap = bool(overlap)  # also register the begin/wait pair -> exercises the chunked pass-T path
        self.calls = 0
        self.elements = 0
        self.waits = 0

    def attach(self, ctx):
        import ctypes as C

        from proximalalgorithms.jl_amd._lib import call

        def fn(ptr, count, pg_dtype, stream):
            self.calls += 1
            self.elements += count
            call("pg_axpby", ctx.handle, pg_dtype, count, C.c_void_p(ptr), float(self.world_size), C.c_void_p(ptr), 0.0,
                 None)

        def wait(stream):
            self.waits += 1

        ctx.set_allreduce(fn)
        if self.overlap:
            ctx.set_allreduce_async(fn, wait)


class ThreadAllReduce:
    """Test double for the collective between SEVERAL CONTEXTS OF ONE PROCESS, one host thread per context (the row-team
    test on a one-GPU box): a SUM all-reduce through the host -- every rank downloads its payload, the ranks meet at a
    barrier, each sums the payloads in rank order and uploads the result.  ``view(rank)`` is what a LeastSquares takes as
    ``comm``."""

    def __init__(self, world_size):
        import threading

        self.world_size = int(world_size)
        self.barrier = threading.Barrier(self.world_size)
        self.slots = [None] * self.world_size
        self.calls = [0] * self.world_size

    def view(self, rank, shard="rows"):
        outer = self
        shard_ = shard

        class _Rank:
            world_size, shard = outer.world_size, shard_

            def attach(self, ctx):
                import ctypes as C

                import numpy as np

                from proximalalgorithms.jl_amd._lib import call
                from proximalalgorithms.jl_amd.device import _PG2NP

                def fn(ptr, count, pg_dtype, stream):
                    outer.calls[rank] += 1
                    host = np.empty(count, _PG2NP[pg_dtype])
                    call("pg_memcpy_d2h", ctx.handle, host.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), host.nbytes)
                    outer.slots[rank] = host
                    outer.barrier.wait(timeout=120)
                    total = outer.slots[0].copy()
                    for q in range(1, outer.world_size):
                        total += outer.slots[q]
                    outer.barrier.wait(timeout=120)  # nobody overwrites a slot before everybody has summed
                    call("pg_memcpy_h2d", ctx.handle, C.c_void_p(ptr), total.ctypes.data_as(C.c_void_p), total.nbytes)

                ctx.set_allreduce(fn)
                ctx.set_column_sharding(outer.world_size if shard_ == "cols" else 0, rank)

        return _Rank()
