"""Component-sharded mixture log-density across GPUs (one process per GPU, RCCL over xGMI).

The reference evaluates the C components in a serial Python loop on one device
(density_experiment.py:562-571); they share nothing but the input batch.  Here rank r
owns a contiguous block of components, computes ``ll_c(x)`` for them in ONE kernel
launch, and a single ``all_gather`` of ``float32[C_local, n]`` per rank rebuilds the
``(C, n)`` table on every rank before the (replicated, cheap) mixture recursion.

No other collective exists on the path.  The exchange is latency-bound (4*n bytes per
component), so consecutive (independent) batches are served in GROUPS: ``GroupPipeline`` runs one flow
launch per group on one HIP stream and the group's all-gather + recursion on a second one,
double-buffered, so the exchange of group g overlaps the kernel of group g+1 (this is the
pipeline ``bench.py`` times and ``tests/test_sharded_gpu.py`` checks on RCCL);
``ShardedMixture.log_prob_pipelined`` is the single-stream form of the same idea.
"""
from __future__ import annotations


def partition(n_components, world_size):
    """Contiguous, explicit component -> rank map; equal blocks (so one all_gather_into_tensor
    rebuilds the (C, n) table in component order).  Returns [(begin, end)] per rank."""
    if n_components % world_size != 0:
        raise ValueError(
            f"{n_components} components do not split evenly over {world_size} ranks; "
            "use a world size that divides the component count (or replicas over batch rows)")
    per = n_components // world_size
    return [(r * per, (r + 1) * per) for r in range(world_size)]


class ShardedMixture:
    """``compute_local(x) -> (C_local, n)`` is the rank's own component launch
    (``NativeMixture.component_log_prob`` over its handles); ``lse(ll, rho) -> (n,)`` is the
    mixture recursion (``native.mixture_lse``).  Both are injected so the collective logic can be
    exercised with ``gloo`` on CPU in tests without any compute path of its own."""

    def __init__(self, n_components, compute_local, lse, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_components = n_components
        self.parts = partition(n_components, self.world)
        self.c_begin, self.c_end = self.parts[self.rank]
        self.compute_local = compute_local
        self.lse = lse

    def gather(self, ll_local, out=None, async_op=False):
        """(C_local, n) on every rank -> (C, n) on every rank, component order preserved."""
        import torch
        c_local, n = ll_local.shape
        if out is None:
            out = torch.empty((self.n_components, n), dtype=ll_local.dtype, device=ll_local.device)
        if self.world == 1:
            out.copy_(ll_local)
            return out, None
        work = self.dist.all_gather_into_tensor(out, ll_local.contiguous(), group=self.group, async_op=async_op)
        return out, work

    def log_prob(self, x, rho):
        ll_local = self.compute_local(x)
        ll, _ = self.gather(ll_local)
        return self.lse(ll, rho), ll

    def log_prob_group(self, xs, rho, compute_local_group):
        """A GROUP of independent batches behind one exchange: ``compute_local_group(xs) -> (C_local, S*n)`` is ONE
        launch over the S batches (NativeMixture.prepared_group_log_prob / gbnf_mixture_component_log_prob_multi;
        batch b in columns [b*n, (b+1)*n)); one all-gather rebuilds the (C, S*n) table, one recursion launch
        finishes all S batches.  Returns [G_0, ..., G_{S-1}]."""
        n = xs[0].shape[0]
        ll_local = compute_local_group(xs)
        if tuple(ll_local.shape) != (self.c_end - self.c_begin, len(xs) * n):
            raise ValueError("compute_local_group must return (C_local, len(xs) * n)")
        ll, _ = self.gather(ll_local)
        G = self.lse(ll, rho)
        return [G[b * n:(b + 1) * n] for b in range(len(xs))]

    def log_prob_pipelined(self, batches, rho):
        """Throughput form: batches are independent, so the all-gather of batch i overlaps the
        flow kernel of batch i+1 (the recursion for batch i is enqueued after kernel i+1)."""
        outs = []
        pending = None
        for x in batches:
            ll_local = self.compute_local(x)
            ll, work = self.gather(ll_local, async_op=True)
            if pending is not None:
                p_ll, p_work = pending
                if p_work is not None:
                    p_work.wait()
                outs.append(self.lse(p_ll, rho))
            pending = (ll, work)
        if pending is not None:
            p_ll, p_work = pending
            if p_work is not None:
                p_work.wait()
            outs.append(self.lse(p_ll, rho))
        return outs


def row_partition(n, world_size):
    """Batch-row replicas: rank r evaluates rows [begin, end) with ALL components.  The alternative to component sharding
    when the component count does not divide over the ranks (e.g. C = 4 on 8 GPUs)."""
    base, rem = divmod(n, world_size)
    out, begin = [], 0
    for r in range(world_size):
        end = begin + base + (1 if r < rem else 0)
        out.append((begin, end))
        begin = end
    return out


class ReplicatedMixture:
    """Every rank holds every component and evaluates its own slice of the batch: NO collective on the data path
    (``log_prob_local``); ``log_prob`` optionally reassembles the (n,) result on every rank with one all-gather of the
    row slices (padded to the largest slice).  ``log_prob_fn(x_rows) -> (rows,)`` is the single-GPU path
    (``BoostedFlow.log_prob`` / ``NativeMixture.log_prob``)."""

    def __init__(self, log_prob_fn, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.log_prob_fn = log_prob_fn

    def log_prob_local(self, x):
        b, e = row_partition(x.shape[0], self.world)[self.rank]
        return self.log_prob_fn(x[b:e]), (b, e)

    def log_prob(self, x):
        import torch
        n = x.shape[0]
        parts = row_partition(n, self.world)
        local, (b, e) = self.log_prob_local(x)
        if self.world == 1:
            return local
        width = max(pe - pb for pb, pe in parts)
        buf = torch.zeros(width, dtype=local.dtype, device=local.device)
        buf[: e - b] = local
        out = torch.empty(self.world * width, dtype=local.dtype, device=local.device)
        self.dist.all_gather_into_tensor(out, buf, group=self.group)
        return torch.cat([out[r * width: r * width + (pe - pb)] for r, (pb, pe) in enumerate(parts)])


class _BoundGroup:
    """A group of input tensors with its launches bound for every buffer slot (GroupPipeline.bind)."""
    __slots__ = ("xs", "launches")

    def __init__(self, xs, launches):
        self.xs, self.launches = xs, launches


class GroupPipeline:
    """The measured multi-GPU pipeline: groups of S independent batches, two HIP streams.

        main stream:  flow kernel of group g     (ONE launch: rank's components x S batches, gbnf_mixture_component_log_prob_multi)
        post stream:  all-gather of the (C_local, S*n) table  ->  (C, S*n)   (RCCL, one collective per group)
                      mixture recursion over the whole table   ->  G (S*n,)  (one launch)

    NBUF = 2 buffer sets: the flow kernel of group g+1 runs while group g is exchanged and finished; a buffer set is
    re-used only after its recursion has completed (event-ordered, no host synchronisation anywhere).  Every launch is
    bound once per distinct group of input tensors (``NativeMixture.prepared_group_log_prob`` /
    ``native.prepared_mixture_lse``): per group the host does two ctypes calls and one RCCL call.

    ``mix``: the rank's ``native.NativeMixture`` (its components [c_begin, c_end) of ``n_components``); ``gather`` False
    (one rank, no collective forced) skips the second stream and the exchange.  ``submit(xs)`` returns the device tensor
    that will hold G of the group's batches (batch b in [b*n, (b+1)*n)) once ``post_done`` of that slot has fired;
    ``drain()`` makes the current stream wait for everything submitted."""

    NBUF = 2

    def __init__(self, mix, n_components, c_begin, c_end, rho, rows, group_size, gather, process_group=None, overlap=True):
        import torch
        from . import native
        self.torch, self.native = torch, native
        self.mix, self.C, self.c0, self.c1 = mix, int(n_components), int(c_begin), int(c_end)
        self.rho, self.rows, self.S = rho, int(rows), int(group_size)
        self.gather = bool(gather)
        self.pg = process_group
        self.dev = rho.device
        self.main = torch.cuda.current_stream(self.dev)
        import ctypes
        self._mptr = ctypes.c_void_p(self.main.cuda_stream)
        # overlap=False: exchange + recursion on the kernel's own stream (a caller with a single group has nothing to overlap
        # them with and saves the cross-stream hand-over)
        self.post = torch.cuda.Stream(self.dev) if (self.gather and overlap) else self.main
        self.flow_done = [torch.cuda.Event() for _ in range(self.NBUF)]
        self.post_done = [torch.cuda.Event() for _ in range(self.NBUF)]
        self._bufs = {}        # group size -> per-slot (local, full, G)
        self._lse = {}         # (group size, slot) -> bound recursion launch (depends on the pipeline's own buffers only)
        self.gi = 0            # groups submitted so far
        self.gather_events = None     # optional [(start, end)] timing events around the all-gathers (bench.py)
        if self.gather:
            import torch.distributed as dist
            self.dist = dist
            if not dist.is_initialized():
                raise RuntimeError("GroupPipeline(gather=True) needs an initialised process group (backend nccl = RCCL)")

    def _buffers(self, size):
        torch = self.torch
        if size not in self._bufs:
            n = size * self.rows
            local = [torch.empty((self.c1 - self.c0, n), dtype=torch.float32, device=self.dev) for _ in range(self.NBUF)]
            full = ([torch.empty((self.C, n), dtype=torch.float32, device=self.dev) for _ in range(self.NBUF)]
                    if self.gather else local)
            G = [torch.empty(n, dtype=torch.float32, device=self.dev) for _ in range(self.NBUF)]
            self._bufs[size] = (local, full, G)
        return self._bufs[size]

    def _launches(self, q, xs):
        """(flow launch, recursion launch) of the group `xs` into buffer slot q.  The flow launch keeps `xs` alive, so it is
        NOT cached here (a caller streaming fresh batches through submit() would pin every input for the pipeline's
        lifetime): ``bind`` hands it to the caller's token, plain ``submit(list)`` builds it per call.  The recursion
        launch only refers to the pipeline's own buffers and is kept."""
        local, full, G = self._buffers(len(xs))
        key = (len(xs), q)
        if key not in self._lse:
            self._lse[key] = self.native.prepared_mixture_lse(full[q], self.rho, G[q])
        return self.mix.prepared_group_log_prob(list(xs), local[q]), self._lse[key]

    def bind(self, xs):
        """Bind the launches of a group of input tensors for every buffer slot ahead of time: ``submit(token)`` then costs
        no per-tensor host work (a caller that replays the same resident batches -- bench.py -- keeps it out of its
        timed region).  The token keeps the tensors alive."""
        if not 1 <= len(xs) <= self.S:
            raise ValueError(f"a group holds 1..{self.S} batches")
        xs = list(xs)
        return _BoundGroup(xs, [self._launches(q, xs) for q in range(self.NBUF)])

    def submit(self, xs, kernel_events=None):
        """Enqueue one group (1..S batches of (rows, d), or a token from ``bind``); ``kernel_events`` = (start, end) timing
        events recorded around the flow launch on the main stream.  Returns (G tensor of the slot, slot index)."""
        import ctypes
        torch = self.torch
        q = self.gi % self.NBUF
        if isinstance(xs, _BoundGroup):
            flow, lse = xs.launches[q]
            xs = xs.xs
        else:
            if not 1 <= len(xs) <= self.S:
                raise ValueError(f"a group holds 1..{self.S} batches")
            flow, lse = self._launches(q, xs)
        local, full, G = self._buffers(len(xs))
        mptr = self._mptr
        two_streams = self.post is not self.main
        if two_streams and self.gi >= self.NBUF:
            self.main.wait_event(self.post_done[q])          # group gi-NBUF has released this buffer set
        if kernel_events is not None:
            kernel_events[0].record(self.main)
        flow(mptr)
        if kernel_events is not None:
            kernel_events[1].record(self.main)
        if self.gather:
            if two_streams:
                self.flow_done[q].record(self.main)
                self.post.wait_event(self.flow_done[q])
            with torch.cuda.stream(self.post):
                ev = None
                if self.gather_events is not None and len(self.gather_events) < 48:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record(self.post)
                self.dist.all_gather_into_tensor(full[q], local[q], group=self.pg)
                if ev is not None:
                    ev[1].record(self.post)
                    self.gather_events.append(ev)
            lse(ctypes.c_void_p(self.post.cuda_stream))
            if two_streams:
                self.post_done[q].record(self.post)
        else:
            lse(mptr)
        self.gi += 1
        return G[q], q

    def drain(self):
        """The main stream waits for every exchange / recursion submitted so far."""
        if self.post is not self.main:
            for ev in self.post_done[: min(self.gi, self.NBUF)]:
                self.main.wait_event(ev)

    def log_prob_groups(self, groups):
        """Convenience form: [[x, ...], ...] -> [G per batch] (cloned out of the slot buffers as each slot completes)."""
        outs = []
        pending = []
        for xs in groups:
            if len(pending) == self.NBUF:      # the slot about to be re-used: copy its result out first (stream-ordered)
                self._collect(pending.pop(0), outs)
            G, q = self.submit(xs)
            pending.append((G, q, len(xs)))
        for item in pending:
            self._collect(item, outs)
        return outs

    def _collect(self, item, outs):
        G, q, k = item
        if self.post is not self.main:
            self.main.wait_event(self.post_done[q])
        n = self.rows
        outs.extend(G[b * n:(b + 1) * n].clone() for b in range(k))


class LibraryGroupPipeline:
    """The group pipeline with the exchange INSIDE the library (round 4; include/gbnf.h, gbnf_mixture_group_log_prob): per group
    the host makes ONE call -- a hipGraphLaunch of {flow launch, repair launch, ncclAllGather, recursion launch} captured once
    per bound group and buffer slot -- instead of two ctypes calls, a torch.distributed collective and the event hand-offs
    between two streams (``GroupPipeline``: ~50 us of host time and ~45 us of hand-off latency per group, as long as the per-rank
    kernel itself at the driver's 20-step invocation).  NBUF = 2 buffer slots, each with its OWN stream and its OWN RCCL
    communicator: consecutive groups are independent, so group g + 1 (slot 1) runs beside group g (slot 0) with no event
    between them, and a slot is re-used in its own stream's order.  ``gather`` False: one rank, no exchange.

    The two slots' collectives run on two communicators of one device with no order between them: that relies on both being
    co-resident (two ncclAllGather kernels of a few workgroups each; true on every run so far, all of them one rank per box --
    a real multi-rank run of this form has not happened yet, which is why ``bench.py`` keeps ``GroupPipeline`` as the default
    for world > 1).

    Same surface as ``GroupPipeline``: ``bind(xs)`` -> token, ``submit(token | xs, kernel_events)`` -> (G tensor of the slot,
    slot), ``drain()``.  The returned G is the slot's LIVE buffer: it is overwritten by the submit after the next one;
    ``self.done[slot]`` is recorded behind the group -- wait on it (or ``drain()``) before reading, and read before re-using
    the slot.  ``graph=False`` keeps the plain library call per group (also the automatic fall-back when a capture is
    refused: ``graph_errors`` lists why)."""

    NBUF = 2

    def __init__(self, mix, n_components, c_begin, c_end, rho, rows, group_size, gather, process_group=None, graph=True, overlap=True):
        import torch
        from . import native
        self.torch, self.native = torch, native
        self.mix, self.C, self.c0, self.c1 = mix, int(n_components), int(c_begin), int(c_end)
        self.rho, self.rows, self.S = rho, int(rows), int(group_size)
        self.gather, self.graph = bool(gather), bool(graph)
        self.dev = rho.device
        self.main = torch.cuda.current_stream(self.dev)
        nslots = self.NBUF if overlap else 1
        self.nslots = nslots
        self.streams = [torch.cuda.Stream(self.dev) for _ in range(nslots)] if overlap else [self.main]
        self.post = self.streams[0]            # (bench.py registers the extra stream(s) it has to wait for)
        self.comms = [native.Comm.from_torch_distributed(process_group) for _ in range(nslots)] if self.gather else [None] * nslots
        if self.gather and (self.c1 - self.c0) * self.comms[0].world != self.C:
            raise ValueError("the component blocks must be equal and contiguous rank by rank (sharded.partition)")
        self.done = [torch.cuda.Event() for _ in range(nslots)]
        self.starts = [torch.cuda.Event() for _ in range(nslots)]
        self._bufs = {}
        self.gi = 0
        self.gather_events = None
        self.graph_errors = []

    def _buffers(self, size):
        torch = self.torch
        if size not in self._bufs:
            n = size * self.rows
            local = [torch.empty((self.c1 - self.c0, n), dtype=torch.float32, device=self.dev) for _ in range(self.nslots)]
            full = ([torch.empty((self.C, n), dtype=torch.float32, device=self.dev) for _ in range(self.nslots)]
                    if self.gather else [None] * self.nslots)
            G = [torch.empty(n, dtype=torch.float32, device=self.dev) for _ in range(self.nslots)]
            self._bufs[size] = (local, full, G)
        return self._bufs[size]

    def bind(self, xs):
        if not 1 <= len(xs) <= self.S:
            raise ValueError(f"a group holds 1..{self.S} batches")
        xs = list(xs)
        local, full, G = self._buffers(len(xs))
        launches = []
        for q in range(self.nslots):
            with self.torch.cuda.device(self.dev):
                gl = self.native.GroupLaunch(self.mix, self.comms[q], xs, self.C, self.rho, local[q], full[q], G[q], graph=self.graph)
            if gl.graph_error:
                self.graph_errors.append(gl.graph_error)
            launches.append(gl)
        return _BoundGroup(xs, launches)

    def submit(self, xs, kernel_events=None):
        import ctypes
        q = self.gi % self.nslots
        token = xs if isinstance(xs, _BoundGroup) else self.bind(xs)
        _, _, G = self._buffers(len(token.xs))
        st = self.streams[q]
        if st is not self.main:
            # every group starts behind whatever the caller has enqueued on the main stream so far (a fresh host-to-device copy of
            # its batches, an update of rho): an event record + wait, ~3 us of host time per group
            ev = self.starts[q]
            ev.record(self.main)
            st.wait_event(ev)
        if kernel_events is not None:
            kernel_events[0].record(st)
        token.launches[q].launch(ctypes.c_void_p(st.cuda_stream))
        if kernel_events is not None:               # (library mode: the events bracket the WHOLE group, exchange and recursion included)
            kernel_events[1].record(st)
        self.done[q].record(st)
        self.gi += 1
        return G[q], q

    def drain(self):
        for q in range(min(self.gi, self.nslots)):
            if self.streams[q] is not self.main:
                self.main.wait_event(self.done[q])
