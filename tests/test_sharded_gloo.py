"""CPU, 2 processes, gloo: the component-sharded mixture path (partition + all-gather + ordering).

The collective logic is exercised with injected per-component log-densities (the product's compute
leg is the HIP launch, which needs a GPU); the oracle checks the gathered result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gbnf_amd import sharded


def test_partition_is_contiguous_and_even():
    assert sharded.partition(8, 1) == [(0, 8)]
    assert sharded.partition(8, 2) == [(0, 4), (4, 8)]
    assert sharded.partition(8, 8) == [(c, c + 1) for c in range(8)]
    with pytest.raises(ValueError):
        sharded.partition(8, 3)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, ll_all, rho, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import gbnf_oracle as oracle
        C, n = ll_all.shape

        def lse(ll, rho_t):      # checker stands in for the HIP recursion kernel
            return torch.from_numpy(oracle.mixture_recursion(ll.numpy(), rho_t.numpy()))

        calls = []

        def compute_local(x):    # this rank's components only
            c0, c1 = sharded.partition(C, world)[rank]
            calls.append((c0, c1))
            return torch.from_numpy(ll_all[c0:c1] + x.numpy()[None, :, 0] * 0.0)

        sm = sharded.ShardedMixture(C, compute_local, lse)
        x = torch.zeros(n, 3)
        G, ll = sm.log_prob(x, torch.from_numpy(rho))
        outs = sm.log_prob_pipelined([x, x, x], torch.from_numpy(rho))

        def compute_group(xs_):   # one "launch" over a group: batch b in columns [b*n, (b+1)*n), scaled to tell them apart
            c0, c1 = sharded.partition(C, world)[rank]
            return torch.cat([torch.from_numpy(ll_all[c0:c1]) - float(b) for b in range(len(xs_))], dim=1)

        grouped = sm.log_prob_group([x, x, x], torch.from_numpy(rho), compute_group)
        # batch-row replicas (no data-path collective): every rank evaluates its rows with all components
        def full_log_prob(rows):
            idx = rows[:, 0].long().numpy()
            return torch.from_numpy(oracle.mixture_recursion(ll_all[:, idx], rho))
        xr = torch.arange(n, dtype=torch.float32).view(-1, 1).repeat(1, 3)      # column 0 carries the row id
        rm = sharded.ReplicatedMixture(full_log_prob)
        local, rows = rm.log_prob_local(xr)
        ret[rank] = (G.numpy(), ll.numpy(), [o.numpy() for o in outs], calls, (sm.c_begin, sm.c_end),
                     [g.numpy() for g in grouped], (rm.log_prob(xr).numpy(), rows, local.numpy()))
    finally:
        dist.destroy_process_group()


def test_two_rank_component_sharding_gloo():
    from oracle import gbnf_oracle as oracle
    rng = np.random.RandomState(0)
    C, n, world = 4, 257, 2
    ll_all = (rng.standard_normal((C, n)) * 5 - 40).astype(np.float32)
    rho = oracle.rho_init(C)
    expect = oracle.mixture_recursion(ll_all, rho)
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, ll_all, rho, ret), nprocs=world, join=True)
    for rank in range(world):
        G, ll, outs, calls, part, grouped, replica = ret[rank]
        rep_G, rep_rows, rep_local = replica       # replicas: own rows only, reassembled result identical on every rank
        assert rep_rows == sharded.row_partition(n, world)[rank] and rep_local.shape == (rep_rows[1] - rep_rows[0],)
        np.testing.assert_array_equal(rep_G, expect)
        assert part == (rank * 2, rank * 2 + 2)
        assert all(c == part for c in calls)                      # each rank computed only its block
        np.testing.assert_array_equal(ll, ll_all)                 # (C, n) rebuilt in component order
        np.testing.assert_allclose(G, expect, rtol=0, atol=0)
        assert len(outs) == 3
        for o in outs:
            np.testing.assert_array_equal(o, expect)
        assert len(grouped) == 3                                  # grouped exchange: batch b == recursion(ll - b)
        for b, gb in enumerate(grouped):
            np.testing.assert_allclose(gb, oracle.mixture_recursion(ll_all - np.float32(b), rho), rtol=0, atol=0)


def test_row_partition_covers_ragged_batches():
    assert sharded.row_partition(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert sharded.row_partition(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]
    assert sharded.row_partition(8, 1) == [(0, 8)]


def _comm_worker(rank, world, port, bad_rank, id_fails, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gbnf_amd import native
        made = []
        if rank == bad_rank:
            native.Comm.probe = staticmethod(lambda: (False, "librccl.so: cannot open shared object file (injected)"))
        else:
            native.Comm.probe = staticmethod(lambda: (True, ""))
        if id_fails:                                       # rank 0's id request fails AFTER a unanimous probe
            def boom():
                raise native.GbnfError("ncclGetUniqueId failed (injected)")
            native.Comm.unique_id = staticmethod(boom)
        else:
            native.Comm.unique_id = staticmethod(lambda: bytes(128))
        native.Comm.__init__ = lambda self, r, w, uid: made.append((r, w, len(uid)))      # (no RCCL here: the hand-shake only)
        try:
            native.Comm.from_torch_distributed()
            ret[rank] = ("ok", made)
        except native.GbnfError as e:
            ret[rank] = ("raised", str(e))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("bad_rank,id_fails", [(1, False), (0, False), (-1, True), (-1, False)])
def test_library_communicator_handshake_never_leaves_a_rank_behind(bad_rank, id_fails):
    """the ranks agree BEFORE any collective of the communicator set-up.  A rank whose local probe fails (RCCL not
    loadable) makes every rank raise; so does an id request that fails on rank 0 after a unanimous probe (rank 0 still performs
    the broadcast, sending None); with neither, every rank reaches ncclCommInitRank with the same 128-byte id.  No case hangs."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_comm_worker, args=(world, _free_port(), bad_rank, id_fails, ret), nprocs=world, join=True)
    if bad_rank >= 0 or id_fails:
        assert all(ret[r][0] == "raised" for r in range(world)), dict(ret)
    else:
        assert [ret[r] for r in range(world)] == [("ok", [(r, world, 128)]) for r in range(world)]
