#!/usr/bin/env python3
"""Per-phase shader-clock times of the LDS-resident DP epilogue kernel for one workgroup under full load (a -DVAEQ_EPI_STAMPS build:
HIPFLAGS=-DVAEQ_EPI_STAMPS tools/build_variant.sh epistamps vae_equalizer_amd/csrc/vaeq_epilogue.hip; VAEQ_LIB=gpurun_variants/libvaeq_epistamps.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vae_equalizer_amd.engine import dp_epilogue_compact
R, N, dev = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 10000, "cuda:0"
amp = (np.arange(-7, 8, 2) / np.sqrt(42.0)).astype(np.float32)
lev = torch.randint(0, 8, (R, 2, 2, N))
data = torch.from_numpy(amp)[lev].to(torch.float16).to(dev)
y = (torch.from_numpy(amp)[lev] + 0.05 * torch.randn(R, 2, 2, N)).to(dev)
eq = y[:, :, 0].contiguous()
dec = lev.to(torch.int8).to(dev)
for _ in range(3):
    e2 = eq.clone()
    res = dp_epilogue_compact(e2, dec, y, data, amp, 0.0, 0.01, 100)
torch.cuda.synchronize()
t = e2[0, 0, :5].cpu().numpy()                       # 100 MHz shader clock counter? (s_memtime: core clock) -> report raw cycles and shares
names = ["pass A (staging + both correlations)", "partial sums + shifts", "SER walk soft demapper (LDS only)", "radius walk (y)", "SER walk constellation (y)"]
tot = t.sum()
for n_, v in zip(names, t):
    print(f"{n_:45s} {v:10.0f} cycles  {100 * v / tot:5.1f} %")
print("total", tot, "cycles; SER", res["SER"][0].cpu().numpy())
