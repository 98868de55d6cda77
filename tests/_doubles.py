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
