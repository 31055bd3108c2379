"""Component-sharded mixture log-density across GPUs (one process per GPU, RCCL over xGMI).

The reference evaluates the C components in a serial Python loop on one device
(density_experiment.py:562-571); they share nothing but the input batch.  Here rank r
owns a contiguous block of components, computes ``ll_c(x)`` for them in ONE kernel
launch, and a single ``all_gather`` of ``float32[C_local, n]`` per rank rebuilds the
``(C, n)`` table on every rank before the (replicated, cheap) mixture recursion.

No other collective exists on the path.  The exchange is latency-bound (4*n bytes per
component), so ``ShardedMixture.log_prob_pipelined`` keeps the all-gather of batch i in
flight under the flow kernel of batch i+1.
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
