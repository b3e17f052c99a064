"""Sweep sharding: run r -> rank r % world, one all_gather at the end.  world_size 2 and 3 over gloo on the CPU."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_runs, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from vae_equalizer_amd import sweep
    r, w, _ = sweep.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    mine = sweep.my_slice(n_runs)
    local = torch.stack([torch.full((3, 5), float(i)) + torch.arange(5.0) for i in mine]) if mine else torch.zeros(0, 3, 5)
    rows = sweep.gather_rows(local, n_runs)
    q.put((rank, mine, rows))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_runs", [(2, 7), (2, 8), (3, 4), (3, 2)])   # (3, 2): one rank owns no run
def test_shard_and_gather(world, n_runs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(k, world, port, n_runs, q)) for k in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = torch.stack([torch.full((3, 5), float(i)) + torch.arange(5.0) for i in range(n_runs)])
    owned = []
    for rank, mine, rows in got:
        assert mine == list(range(rank, n_runs, world))
        assert torch.equal(rows, expect)          # every rank ends with the full table, in run order
        owned += mine
    assert sorted(owned) == list(range(n_runs))   # a partition: every run trained exactly once


def test_single_process_is_identity():
    from vae_equalizer_amd import sweep
    assert sweep.my_slice(5, 0, 1) == [0, 1, 2, 3, 4]
    x = torch.randn(5, 2)
    assert sweep.gather_rows(x, 5, 0, 1) is x


def _worker_forced(port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", VAEQ_FORCE_COLLECTIVE="1")
    from vae_equalizer_amd import sweep
    r, w, _ = sweep.init_distributed(backend="gloo")            # VAEQ_FORCE_COLLECTIVE: a process group even at world size 1
    x = torch.arange(35.0).reshape(7, 5)
    rows = sweep.gather_rows(x, 7)                                # ... and the collective branch instead of the identity shortcut
    q.put((r, w, dist.is_initialized(), rows is x, rows))
    dist.destroy_process_group()


def test_forced_collective_at_world_size_one():
    """What tests/test_bench_gpu.py::test_rccl_gather_single_rank runs with backend nccl on the GPU box, rehearsed over gloo."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_forced, args=(_free_port(), q))
    p.start()
    r, w, init, same, rows = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0 and (r, w) == (0, 1) and init and not same
    assert torch.equal(rows, torch.arange(35.0).reshape(7, 5))
