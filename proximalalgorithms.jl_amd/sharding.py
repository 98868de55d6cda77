"""Sharding of A over the GPUs of one node, one process per GPU.

Row sharding (``shard="rows"``): rows of A and entries of b are the independent units; every gradient evaluation ends
with ONE sum all-reduce of [grad ; f] (n+1 elements) over RCCL/xGMI, every f-only evaluation with a 1-element
all-reduce.  All n-vectors are replicated, so the elementwise epilogue and its reductions need no communication.

Column sharding (``shard="cols"``): every rank holds a column block of A and the matching slices of the n-vectors; b and
the residual are replicated.  A' r is then local and what crosses ranks is A x (m elements) plus 8 * world scalar slots (four scalars per rank, each as a hi / lo pair)
-- ONE all-reduce per iteration, 64x smaller than the row-sharded payload at the headline shape -- so every rank keeps
the single-sweep iteration (A read once per iteration).

Row teams (``attach_row_team`` / ``row_team_in_process``): the row layout at ONE read of A per iteration -- the GPUs exchange
the per-column partial dots inside the sweep kernel through each other's inbox (peer-visible memory over xGMI, mapped with
IPC handles; csrc/pg_gemv_tn4.hip) instead of all-reducing A' r between two sweeps.
"""

def shard_rows(m_global, world_size, rank):
    """Contiguous, balanced row partition: returns (row_offset, m_local)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, rem = divmod(int(m_global), int(world_size))
    m_local = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, m_local


def shard_cols(n_global, world_size, rank):
    """Contiguous, balanced column partition: returns (col_offset, n_local)."""
    return shard_rows(n_global, world_size, rank)


def allreduce_sum_(tensor, group=None):
    """In-place SUM all-reduce of a torch tensor over the process group (RCCL on GPUs, gloo on CPU)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
    return tensor


class TorchDistributedComm:
    """All-reduce provider backed by torch.distributed (backend "nccl" == RCCL on ROCm).  ``attach(ctx)``
    registers the C callback the library invokes between the local GEMV passes and the replicated epilogue;
    the collective is enqueued on the context's stream (torch's current stream), so no host sync is added."""

    def __init__(self, group=None, overlap=False, shard="rows"):
        import torch.distributed as dist

        if shard not in ("rows", "cols"):
            raise ValueError("shard must be 'rows' or 'cols'")
        self.shard = shard

        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.overlap = bool(overlap)  # pipeline the [grad ; f] all-reduce with pass T (column chunks)
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._views = {}
        self.calls = 0
        self.elements = 0

    def _view(self, ctx, ptr, count, pg_dtype):
        import torch

        from .device import _PG2NP, _RawDeviceArray

        key = (ptr, count, pg_dtype)
        t = self._views.get(key)
        if t is None:
            t = torch.as_tensor(_RawDeviceArray(ptr, count, _PG2NP[pg_dtype], None), device=ctx.torch_device)
            self._views[key] = t
        return t

    def attach(self, ctx):
        import torch.distributed as dist

        import torch

        def on_ctx_stream():
            """the collective must be ordered with the library's kernels: when the context runs on its own stream (not
            torch's current one), make that stream current for the call"""
            if ctx.stream and ctx.stream != torch.cuda.current_stream(ctx.torch_device).cuda_stream:
                return torch.cuda.stream(torch.cuda.ExternalStream(ctx.stream, device=ctx.torch_device))
            import contextlib

            return contextlib.nullcontext()

        def fn(ptr, count, pg_dtype, stream):
            self.calls += 1
            self.elements += count
            with on_ctx_stream():
                allreduce_sum_(self._view(ctx, ptr, count, pg_dtype), self.group)

        pending = []

        def begin(ptr, count, pg_dtype, stream):
            # async_op=True: RCCL runs the collective on its own stream behind an event recorded on the current
            # stream; the current stream is NOT made to wait until wait() -> the next pass-T chunk overlaps it
            self.calls += 1
            self.elements += count
            with on_ctx_stream():
                pending.append(dist.all_reduce(self._view(ctx, ptr, count, pg_dtype), op=dist.ReduceOp.SUM,
                                               group=self.group, async_op=True))

        def wait(stream):
            with on_ctx_stream():
                for w in pending:
                    w.wait()
            pending.clear()

        ctx.set_allreduce(fn)
        if self.overlap and self.shard == "rows":
            ctx.set_allreduce_async(begin, wait)
        ctx.set_column_sharding(self.world_size if self.shard == "cols" else 0, self.rank)


def native_rccl_available():
    """True when the library can bind RCCL itself (dlopen librccl): the collective then needs no host callback."""
    from ._lib import load

    return bool(load().pg_comm_available())


class NativeRcclComm:
    """The library's own RCCL communicator (csrc/pg_comm.hip): no Python in the collective path.  The 128-byte
    ncclUniqueId is created on rank 0 and shipped with torch.distributed (any backend) when world_size > 1.  A context
    keeps ONE communicator for the life of the job: further NativeRcclComm objects attached to it reuse that communicator
    and only switch the layout (rows / cols) and the overlap mode."""

    def __init__(self, world_size=None, rank=None, overlap=False, shard="rows"):
        import torch.distributed as dist

        if shard not in ("rows", "cols"):
            raise ValueError("shard must be 'rows' or 'cols'")
        self.shard = shard

        if world_size is None:
            initialised = dist.is_available() and dist.is_initialized()
            world_size = dist.get_world_size() if initialised else 1
            rank = dist.get_rank() if initialised else 0
        self.world_size, self.rank, self.overlap = int(world_size), int(rank), bool(overlap)
        self._ctx = None
        self._calls0 = self._elements0 = 0

    def _stats(self):
        import ctypes as C

        from ._lib import call

        if self._ctx is None:
            return 0, 0
        calls, elements = C.c_int64(0), C.c_int64(0)
        call("pg_ctx_comm_stats", self._ctx.handle, C.byref(calls), C.byref(elements))
        return calls.value - self._calls0, elements.value - self._elements0

    @property
    def calls(self):
        """all-reduces issued through this communicator since it was attached"""
        return self._stats()[0]

    @property
    def elements(self):
        return self._stats()[1]

    def attach(self, ctx):
        import ctypes as C

        import torch.distributed as dist

        from ._lib import call

        if self._ctx is ctx:
            return
        if self._ctx is not None:
            raise RuntimeError("a NativeRcclComm is bound to one context")
        ident = C.create_string_buffer(128)
        shape = getattr(ctx, "_native_comm_shape", None)
        if shape is None:
            if self.rank == 0:
                call("pg_comm_get_unique_id", ident)
            if self.world_size > 1:
                box = [bytes(ident.raw) if self.rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                ident = C.create_string_buffer(box[0], 128)
        elif shape != (self.world_size, self.rank):
            raise RuntimeError("the context already has a communicator for world_size=%d rank=%d" % shape)
        call("pg_ctx_comm_init", ctx.handle, ident, self.world_size, self.rank,
             1 if (self.overlap and self.shard == "rows") else 0)
        ctx._native_comm_shape = (self.world_size, self.rank)
        ctx.set_column_sharding(self.world_size if self.shard == "cols" else 0, self.rank)
        self._ctx = ctx
        self._calls0, self._elements0 = 0, 0
        self._calls0, self._elements0 = self._stats()


def _row_team_alloc(ctx):
    import ctypes as C

    from ._lib import call

    inbox, nbytes = C.c_void_p(), C.c_int64()
    call("pg_ctx_row_team_alloc", ctx.handle, C.byref(inbox), C.byref(nbytes))
    return inbox.value


def _row_team_set(ctx, rank, inboxes, max_workgroups):
    import ctypes as C

    from ._lib import call

    arr = (C.c_void_p * len(inboxes))(*inboxes)
    call("pg_ctx_set_row_team", ctx.handle, len(inboxes), int(rank), arr, int(max_workgroups))
    ctx._row_team = (len(inboxes), int(rank), int(max_workgroups))


def _row_team_selftest(ctx, world_size):
    """one scalar exchange through the freshly mapped inboxes (every rank calls it at the same point): 'ok' when the sum of the
    ranks' contributions came back, else what went wrong -- the sweeps still run (they fall back to two reads when a peer's
    granules do not arrive), but the record says why"""
    import ctypes as C

    from ._lib import ProxGradError, call

    got = C.c_double()
    try:
        call("pg_ctx_row_team_selftest", ctx.handle, C.byref(got))
    except ProxGradError as e:
        return "failed: %s" % str(e)[:160]
    want = world_size * (world_size + 1) / 2
    return "ok" if got.value == want else "wrong sum %r (expected %r)" % (got.value, want)


def attach_row_team(ctx, world_size=None, rank=None, group=None, max_workgroups=0):
    """One process per GPU: make the row-sharded job on ``ctx`` a row team (pg_ctx_set_row_team).  Every rank allocates its
    inbox, the IPC handles travel with torch.distributed (any backend), every rank opens its peers' inboxes.  The collective
    registered on the context (TorchDistributedComm / NativeRcclComm, shard="rows") stays in place for initialisation, the
    line search and the two-sweep fallback.  Call on every rank at the same point of the program."""
    import ctypes as C

    import torch.distributed as dist

    from ._lib import call

    if world_size is None:
        world_size, rank = dist.get_world_size(group), dist.get_rank(group)
    if world_size <= 1:
        call("pg_ctx_set_row_team", ctx.handle, 0, 0, None, 0)
        return
    env_knobs = row_team_knobs_from_env()
    if env_knobs:  # PG_ROW_TEAM_TUNE="PAIR=1,SPIN=4194304": the sweep's knobs for a whole job, every rank the same environment
        row_team_tune(ctx, **env_knobs)
    cached = getattr(ctx, "_row_team_inboxes", None)
    if cached is not None and cached[0] == (world_size, rank):  # the peers' inboxes are mapped once per context
        _row_team_set(ctx, rank, cached[1], max_workgroups)
        dist.barrier(group=group)
        ctx._row_team_selftest = _row_team_selftest(ctx, world_size)
        return
    own = _row_team_alloc(ctx)
    handle = C.create_string_buffer(64)
    call("pg_ctx_row_team_export", ctx.handle, handle)
    handles = [None] * world_size
    dist.all_gather_object(handles, bytes(handle.raw), group=group)
    inboxes = []
    for q, h in enumerate(handles):
        if q == rank:
            inboxes.append(own)
        else:
            p = C.c_void_p()
            call("pg_ctx_row_team_import", ctx.handle, C.create_string_buffer(h, 64), C.byref(p))
            inboxes.append(p.value)
    _row_team_set(ctx, rank, inboxes, max_workgroups)
    ctx._row_team_inboxes = ((world_size, rank), inboxes)
    dist.barrier(group=group)  # nobody sweeps before every inbox is mapped and zeroed
    ctx._row_team_selftest = _row_team_selftest(ctx, world_size)


def row_team_in_process(contexts, max_workgroups=0):
    """Several contexts of ONE process as a row team (plain device pointers, no IPC): contexts[p] plays device p.  On a
    one-GPU box this runs the whole protocol between streams of the same device, each context limited to
    ``max_workgroups`` workgroups so that all members are resident together (tests; a functional check, not a layout)."""
    inboxes = [_row_team_alloc(c) for c in contexts]
    for p, c in enumerate(contexts):
        _row_team_set(c, p, inboxes, max_workgroups)


ROW_TEAM_KNOBS = ("C", "LAG", "LAGR", "PF", "WGS", "W", "K1", "PAIR", "AHEAD", "SPIN")


def row_team_knobs_from_env(text=None):
    """{knob: value} of the environment variable PG_ROW_TEAM_TUNE ("PAIR=1,LAG=2,SPIN=4194304"; read by attach_row_team on every rank --
    a job's launcher exports it to all of them): the run-time form of pg_ctx_row_team_tune, no rebuild and no PG_TUNE."""
    import os

    text = os.environ.get("PG_ROW_TEAM_TUNE", "") if text is None else text
    out = {}
    for part in filter(None, (p.strip() for p in text.replace(";", ",").split(","))):
        key, _, value = part.partition("=")
        if key.strip() not in ROW_TEAM_KNOBS or not value.strip().isdigit():
            raise ValueError("PG_ROW_TEAM_TUNE: %r is not <knob>=<non-negative integer> with a knob of %s" % (part, ", ".join(ROW_TEAM_KNOBS)))
        out[key.strip()] = int(value)
    return out


def row_team_tune(ctx, **knobs):
    """The row-team sweep's geometry for this context, at run time (pg_ctx_row_team_tune; every rank the same values, 0 = the
    library's choice): row_team_tune(ctx, PAIR=1) -- one post per two steps, half the fabric transactions; SPIN=... -- the bounded
    wait in polls; C / LAG / LAGR (value - 1 tiles in registers) / PF / WGS / W / K1.  Takes effect at the next sweep."""
    from ._lib import call

    for key, value in knobs.items():
        if key not in ROW_TEAM_KNOBS:
            raise ValueError("unknown row-team knob %r (one of %s)" % (key, ", ".join(ROW_TEAM_KNOBS)))
        call("pg_ctx_row_team_tune", ctx.handle, key.encode(), int(value))


def row_team_geometry(ctx):
    """what the LAST row-team sweep of this context ran with, as the library prints it ("W=1 U=8 C=2 LAG=2 LAGR=2 PF=2 WGS=4 K1=1
    PAIR=0 SPIN=2097152 WG=1024"; "none" before the first one) and as a dict"""
    import ctypes as C

    from ._lib import call

    buf = C.create_string_buffer(160)
    call("pg_ctx_row_team_geometry", ctx.handle, buf, len(buf))
    text = buf.value.decode()
    return text, ({k: int(v) for k, v in (kv.split("=") for kv in text.split())} if text != "none" else {})


def row_team_stats(ctx):
    """{sweeps, late_waves, wait_polls} since the context became a row team (pg_ctx_row_team_stats): how often a wave did not
    find a step's granules at its first look, and how long it polled -- the first thing to read after a run on real fabric"""
    import ctypes as C

    from ._lib import call

    a, b, c_ = C.c_int64(), C.c_int64(), C.c_int64()
    call("pg_ctx_row_team_stats", ctx.handle, C.byref(a), C.byref(b), C.byref(c_))
    return {"sweeps": a.value, "late_waves": b.value, "wait_polls": c_.value}
