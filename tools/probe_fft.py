#!/usr/bin/env python3
"""hipFFT (torch.fft) time of the generator's batched c2c transform [2048, 2, L] for candidate row lengths L."""
import sys, torch
dev = "cuda:0"
for L in [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "20034,20250,20480,20736,21600,24576,25000,32768".split(","))]:
    x = torch.randn(2048, 2, L, dtype=torch.complex64, device=dev)
    for _ in range(2):
        y = torch.fft.ifft(torch.fft.fft(x, dim=-1), dim=-1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        y = torch.fft.ifft(torch.fft.fft(x, dim=-1), dim=-1)
    e1.record(); torch.cuda.synchronize()
    print(f"L={L:6d}: fft+ifft {e0.elapsed_time(e1) / 5:7.3f} ms  ({2 * 2 * x.numel() * 8 / (e0.elapsed_time(e1) / 5) / 1e6:6.0f} GB/s r+w)", flush=True)
    del x, y
