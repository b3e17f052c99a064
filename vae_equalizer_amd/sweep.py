"""Sharding of an embarrassingly parallel parameter sweep over the GPUs of one node.

Run index r goes to rank r % world_size; every rank trains its runs with no communication, then ONE all_gather of
the per-run result rows (RCCL over xGMI when the backend is ``nccl``; ``gloo`` in the CPU tests) reassembles the
full result tensors on every rank.
"""
import os

import torch
import torch.distributed as dist


def dist_info():
    """(rank, world_size, local_rank) from torch.distributed if initialised, else from the launcher's environment."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size(), int(os.environ.get("LOCAL_RANK", 0))
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init_distributed(backend=None):
    """Initialise the process group when launched by torch.distributed.run with WORLD_SIZE > 1 (one process per GPU).
    VAEQ_FORCE_COLLECTIVE=1 initialises it at WORLD_SIZE = 1 as well (under torch.distributed.run --nproc-per-node 1), so that RCCL's
    initialisation and the gather's collective run on a one-GPU box exactly as they do on N GPUs (gather_rows)."""
    world = int(os.environ.get("WORLD_SIZE", 1))
    force = bool(os.environ.get("VAEQ_FORCE_COLLECTIVE")) and "RANK" in os.environ
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = os.environ.get("VAEQ_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        dist.init_process_group(backend=backend)
    return dist_info()


def device_for_rank(local_rank, world):
    """cuda:LOCAL_RANK (one process per GPU); VAEQ_SINGLE_DEVICE=1 maps every rank to cuda:0 to rehearse N > 1 on a one-GPU box."""
    if os.environ.get("VAEQ_SINGLE_DEVICE") or os.environ.get("VAEQ_BENCH_SINGLE_DEVICE"):
        return torch.device("cuda", 0)
    return torch.device("cuda", local_rank if world > 1 else torch.cuda.current_device())


def stream_seed(base_seed, rank, batch_index=0):
    """Key of a rank's device-generator streams for one batch of runs: reproducible from ``base_seed``, fresh entropy per invocation when it
    is None (the reference seeds nothing), and never shared between ranks or between the batches (problem shapes) of one rank."""
    if base_seed is None:
        import numpy as np
        base_seed = int(np.random.SeedSequence().entropy & 0xFFFFFFFFFFFF)
    return int(base_seed) + 7919 * int(rank) + 104729 * int(batch_index)


def my_slice(n_runs, rank=None, world=None):
    """Indices of the runs this rank owns: r = rank (mod world)."""
    if rank is None:
        rank, world, _ = dist_info()
    return list(range(rank, n_runs, world))


def gather_rows(local_rows, n_runs, rank=None, world=None, force_collective=None):
    """local_rows[len(my_slice), ...] float32 -> rows[n_runs, ...] in run order, on every rank (a single all_gather).

    force_collective (default: the environment's VAEQ_FORCE_COLLECTIVE): at world == 1 still go through the process group's
    all_gather_into_tensor on device tensors when one is initialised -- the very code N ranks execute, so RCCL (backend ``nccl``) can be
    exercised on a single GPU (tests/test_bench_gpu.py::test_rccl_gather_single_rank)."""
    if rank is None:
        rank, world, _ = dist_info()
    if force_collective is None:
        force_collective = bool(os.environ.get("VAEQ_FORCE_COLLECTIVE"))
    if world == 1 and not (force_collective and dist.is_available() and dist.is_initialized()):
        return local_rows
    per = (n_runs + world - 1) // world
    shape = (per,) + tuple(local_rows.shape[1:])
    backend = dist.get_backend()
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    buf = torch.zeros(shape, dtype=torch.float32, device=dev)
    buf[:local_rows.shape[0]] = local_rows.to(dev)
    allbuf = torch.empty((world * per,) + shape[1:], dtype=torch.float32, device=dev)   # concatenation form: nccl and gloo
    dist.all_gather_into_tensor(allbuf, buf)
    allbuf = allbuf.cpu().reshape((world,) + shape)
    out = torch.empty((n_runs,) + tuple(local_rows.shape[1:]), dtype=torch.float32)
    for k in range(world):
        out[k::world] = allbuf[k, :len(range(k, n_runs, world))]                    # run r lives on rank r % world
    return out
