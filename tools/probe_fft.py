#!/usr/bin/env python3
"""hipFFT (torch.fft) time of the generator's batched c2c transform [2048, 2, L]: candidate row lengths and call variants."""
import sys, torch
dev = "cuda:0"


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


if len(sys.argv) > 1 and sys.argv[1] == "variants":
    L = 20480
    x = torch.randn(2048, 2, L, dtype=torch.complex64, device=dev)
    print("fft  new out      ", timed(lambda: torch.fft.fft(x, dim=-1)))
    print("fft  out=x        ", timed(lambda: torch.fft.fft(x, dim=-1, out=x)))
    y = torch.empty_like(x)
    print("fft  out=y        ", timed(lambda: torch.fft.fft(x, dim=-1, out=y)))
    print("ifft backward new ", timed(lambda: torch.fft.ifft(x, dim=-1)))
    print("ifft forward  new ", timed(lambda: torch.fft.ifft(x, dim=-1, norm="forward")))
    print("ifft forward out=x", timed(lambda: torch.fft.ifft(x, dim=-1, norm="forward", out=x)))
    print("ifft forward out=y", timed(lambda: torch.fft.ifft(x, dim=-1, norm="forward", out=y)))
    print("fft(norm=forward) new (scaled fwd)", timed(lambda: torch.fft.fft(x, dim=-1, norm="forward")))
    sys.exit(0)
for L in [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "20034,20250,20480,20736,21600,24576,25000,32768".split(","))]:
    x = torch.randn(2048, 2, L, dtype=torch.complex64, device=dev)
    ms = timed(lambda: torch.fft.ifft(torch.fft.fft(x, dim=-1), dim=-1))
    print(f"L={L:6d}: fft+ifft {ms:7.3f} ms  ({2 * 2 * x.numel() * 8 / ms / 1e6:6.0f} GB/s r+w)", flush=True)
    del x
