#!/usr/bin/env python3
"""Headline benchmark: equalized DP symbols/s on optical DP 64-QAM VAE-LE (BASELINE.json metric, SURVEY config 3).

One bench "step" = one frame of the reference's training loop (N_frame_max = 10 000 DP symbols = 100 minibatch
steps of 100 symbols: FIR + soft demap + ELBO + backward + Adam each, optical_DP_channel/func_VAELE_DP_MQAM_shaping.py:57-66)
for every one of the R independent runs of this GPU's shard of the sweep -- ONE launch of the fused HIP kernel.
Config 3's own sweep is 3 learning rates x iter=5 = 15 runs (Eval_run_DP.py:41,44), which cannot fill a 256-CU GPU;
the workload keeps every other constant of config 3 and raises the seed axis (``iter``) so that R runs per GPU saturate it
(SURVEY 8d "saturation variant").  Received samples come from the on-device channel simulator (synthetic, distinct per run and
per frame) and are resident in HBM before the timed region starts.

N > 1: one process per GPU (torch.distributed.run), each rank owns its own R runs (weak scaling, the sweep is
embarrassingly parallel), no data-path collective; the only communication is the final all_gather of the per-run result
rows, inside the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_DP_SYMBOL = 176      # 32 B rx read + 128 B q write + 16 B out write (SURVEY 8d, 64-QAM, 2 sps)
HBM_PEAK_GBS = 8000.0               # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

CFG = dict(mod="64-QAM", sps=2, nu=0.0, channel="h0", SNR=23.0, symb_rate=90e9, tau_cd=-26e-24, tau_pmd=0.1e-12 * np.sqrt(1000),
           phiIQ=np.array([0.0314, 0.0314], dtype=np.complex64), theta=np.pi / 10, theta_diff=0.06 * np.pi, M_est=25, batch_len=100,
           N_frame_max=10000, lr_optim_vec=[2.5e-3, 2e-3, 3e-3])


def make_frames(n_frames, R, device, seed):
    """rx for n_frames frames x R runs from the on-device channel simulator: list of [R,1,2,2,S] tensors."""
    from vae_equalizer_amd import channel as ch
    from vae_equalizer_amd import shared_funcs as sfun
    t = sfun.qam_tables(CFG["mod"], CFG["nu"])
    h_ch = sfun.upsampled_channel(CFG["channel"], CFG["sps"])
    frames = []
    for f in range(n_frames):       # HIP generator kernels + hipFFT (vaeq_gen_dp_*), Philox streams keyed by (seed, frame, run)
        rx, _ = ch.generate_batch_hip(R, CFG["N_frame_max"], t["amps"], t["P"], CFG["SNR"], h_ch, CFG["symb_rate"], CFG["sps"], CFG["tau_cd"],
                                      CFG["tau_pmd"], CFG["phiIQ"], CFG["theta"] + f * CFG["theta_diff"], device, seed, f)
        frames.append(rx.unsqueeze(1))
    return frames, t


def host_cores():
    """CPU cores this process may actually use: affinity mask, capped by the cgroup CPU quota (a 1-GPU box of the pool
    owns 16 of the host's cores) -- never os.cpu_count() of the whole host."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    if os.environ.get("VAEQ_CPU_THREADS"):
        n = int(os.environ["VAEQ_CPU_THREADS"])
    return min(n, 16) if n > 64 else n


def cpu_baseline(frame_rx, t, var, lr, target_s, threads):
    """The C oracle (oracle/, a port of the reference's step) on the host cores, OpenMP over runs, on a bounded sample of
    the same workload: the first `Rc` runs of the first frame, trained repeatedly until ~target_s of wall time."""
    import oracle
    cores = threads or host_cores()
    Rc = min(frame_rx.shape[0], 8 * cores)
    rx = frame_rx[:Rc, 0].cpu().numpy().copy()
    M, B, sps, n = CFG["M_est"], CFG["batch_len"], CFG["sps"], len(t["amps"])
    steps = CFG["N_frame_max"] // B
    W = np.zeros((Rc, 2, 4, M), np.float32)
    h = np.zeros((Rc, 2, 2, 2, M), np.float32)
    W[:, 0, 0, M // 2] = W[:, 1, 1, M // 2] = 1
    h[:, 0, 0, 0, M // 2] = h[:, 1, 1, 0, M // 2] = 1
    mW, vW, mh, vh = np.zeros_like(W), np.zeros_like(W), np.zeros_like(h), np.zeros_like(h)
    step = np.zeros(Rc, np.int32)
    amp = t["amps"].astype(np.float32)
    P = np.tile(t["P"].astype(np.float32), (Rc, 1))
    varr = np.full((Rc, 2), var, np.float32)
    nu = np.full(Rc, t["nu_sc"], np.float32)
    lrs = np.asarray(lr[:Rc], np.float32)
    q = np.zeros((Rc, 2, 2 * n, steps * B), np.float32)
    y = np.zeros((Rc, 2, 2, steps * B), np.float32)
    loss, ve = np.zeros((Rc, steps), np.float32), np.zeros((Rc, 2, steps), np.float32)
    done, t0, used = 0, time.perf_counter(), cores
    while True:
        used = oracle.dp_train_batch_f32(Rc, cores, steps, B, sps, M, n, B, 0, B, rx, W, h, mW, vW, mh, vh, step, amp, P, varr, nu, lrs,
                                         lrs, q, y, loss, ve)
        done += 1
        el = time.perf_counter() - t0
        if el >= target_s or done >= 200:
            break
    sym = done * Rc * steps * B
    return {"value": sym / el, "unit": "DP-symbols/s", "cores": int(used), "kind": "port",
            "sample": f"{Rc} runs x {done} frame(s) x {steps * B} DP symbols of the same config, fp32 C oracle with OpenMP over runs, "
                      f"{el:.1f} s wall; per core {sym / el / used:.0f} DP-symbols/s"}, loss[:, -1].copy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--runs", type=int, default=0, help="independent runs per GPU (one workgroup each); 0 = 4 x the number of runs "
                    "the device keeps co-resident (vaeq_dp_resident_runs), i.e. four full rounds, no ragged tail")
    ap.add_argument("--threads", type=int, default=0, help="workgroup size per run (0 = library default)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--distinct-frames", type=int, default=4, help="distinct synthetic frames cycled through the steps")
    args = ap.parse_args()

    from vae_equalizer_amd import sweep
    from vae_equalizer_amd.engine import DPEngine
    rank, world, local_rank = sweep.init_distributed(os.environ.get("VAEQ_DIST_BACKEND"))   # default nccl (= RCCL); gloo only to
    if os.environ.get("VAEQ_BENCH_SINGLE_DEVICE"):                                          # rehearse N > 1 on a one-GPU box
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    import torch.distributed as dist
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    K, Wm = args.steps, args.warmup
    B, M, sps = CFG["batch_len"], CFG["M_est"], CFG["sps"]
    from vae_equalizer_amd import _native as nat
    resident = int(nat.lib().vaeq_dp_resident_runs(B, sps, M, 8, args.threads))
    if resident <= 0:
        raise SystemExit(f"vaeq_dp_resident_runs failed: {resident}")
    R = args.runs if args.runs > 0 else 4 * resident
    steps_per_frame = CFG["N_frame_max"] // B
    n_distinct = max(1, min(args.distinct_frames, K + Wm))
    frames, t = make_frames(n_distinct, R, device, seed=1000 + rank)
    var = t["pow_mean"] / 10 ** (CFG["SNR"] / 10) / 2
    lr = np.array([CFG["lr_optim_vec"][(rank * R + i) % 3] for i in range(R)], np.float32)     # the sweep's lr axis
    eng = DPEngine(R, M, t["amps"], t["P"], [var, var], t["nu_sc"], device, sps, args.threads)
    lr_t = torch.tensor(lr, device=device)

    def one_step(k):
        return eng.train(frames[k % n_distinct], B, steps_per_frame, lr_t)

    # allocator priming (untimed setup, not a warm-up step of the metric): two sets of output buffers are alive at the hand-over
    # `out = one_step(...)`, so both are put into the caching allocator now and no hipMalloc lands in the timed region even for W <= 1
    _prime = [torch.empty(R, 1, 2, c, steps_per_frame * B, dtype=torch.float32, device=device) for c in (16, 16, 2, 2)]
    del _prime
    for k in range(Wm):
        out = one_step(k)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        ev[k][0].record()          # HIP events on the stream the kernel is launched on (torch's current stream)
        out = one_step(Wm + k)
        ev[k][1].record()
    rows = torch.cat([out["loss"][:, 0, -1:], out["var_est"][:, 0, :, -1]], dim=1)             # per-run result row
    if world > 1:
        allrows = sweep.gather_rows(rows, world * R, rank, world)                              # the sweep's single gather (RCCL over xGMI)
        assert allrows.shape[0] == world * R
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([el], device=device if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    assert torch.isfinite(rows).all(), "non-finite training result"

    if rank == 0:
        sym_per_launch = R * CFG["N_frame_max"]
        value = world * sym_per_launch * K / el
        achieved = ALGO_BYTES_PER_DP_SYMBOL * sym_per_launch / (kern_ms * 1e-3) / 1e9
        traffic = None
        tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tj):
            try:
                d = json.load(open(tj))
                if d.get("runs") == R and d.get("threads", 0) == args.threads:  # same launch geometry as this run
                    traffic = d.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        res = {
            "metric": "equalized symbols/s/GPU, DP 64-QAM VAE-LE", "value": value, "unit": "DP-symbols/s (1 DP symbol = 2 polarisation symbols)",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": el / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (on-device DP channel simulator vaeq_gen_dp_*: Philox PCS draw, RRC, CD+PMD+rotation via hipFFT, AWGN)",
            "per_gpu": value / world,
            "config": {"workload": "SURVEY config 3: optical DP 64-QAM VAE-LE, nu=0, SNR 23 dB, h0, 90 GBd, M_est=25, batch_len=100, "
                                   "N_frame_max=10000 (100 minibatch steps per bench step), lr in {2.5e-3,2e-3,3e-3}; seed axis raised to "
                                   f"{R} independent runs per GPU (script default iter=5 -> 15 runs)",
                       "runs_per_gpu": R, "resident_runs_per_gpu": resident, "dp_symbols_per_step_per_gpu": sym_per_launch,
                       "kernel_choice": args.threads,
                       "distinct_frames": n_distinct, "parallelism": f"sweep-sharded x{world}, one all_gather of result rows"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "vaeq::dp_wave_kernel<25,8,100,true,1,1>" if args.threads in (0, 1) else "vaeq::dp_train_kernel", "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_DP_SYMBOL * sym_per_launch},
        }
        if not args.no_cpu_baseline and world == 1:      # reported baseline: rank 0 at N = 1 only
            cb, cpu_loss = cpu_baseline(frames[0], t, var, lr, args.cpu_seconds, 0)
            res["cpu_baseline"] = cb
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
