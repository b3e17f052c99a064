#!/usr/bin/env python3
"""RCCL sanity on a one-GPU box: a world-size-1 process group on backend "nccl" runs the same collectives the sweep uses
(barrier, all_gather_into_tensor in concatenation form, all_reduce MAX for the bench clock)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
t0 = time.time()
dist.init_process_group(backend="nccl", rank=0, world_size=1)
x = torch.arange(12, dtype=torch.float32, device="cuda").reshape(3, 4)
out = torch.empty(3, 4, device="cuda")
dist.all_gather_into_tensor(out, x)
t = torch.tensor([1.5], device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier(); torch.cuda.synchronize()
assert torch.equal(out, x) and float(t) == 1.5
print("nccl world-1 ok, backend", dist.get_backend(), f"{time.time() - t0:.1f} s")
dist.destroy_process_group()
