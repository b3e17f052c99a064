"""Batched driver of the DP VAE-LE / VAEflex Monte-Carlo runs: what ``processing()`` of
optical_DP_channel/func_VAELE_DP_MQAM_shaping.py:17-95 and func_VAEflex_DP_MQAM_shaping.py:16-90 does for one run,
done for R runs per frame with one training-kernel launch (engine.DPEngine) and one epilogue-kernel launch (engine.dp_epilogue).

Runs in one batch share (mod, sps, M_est, batch_len, N_frame_max, num_frames, flex_step, channel, N_lrhalf) -- the
shape of the problem -- and may differ in SNR, nu, theta_diff, theta, lr_optim and seed; with the host generator ("numpy") also in
symb_rate (the on-device generators simulate one symbol rate per call: Eval_run_DP batches by symbol rate, mixed batches are refused).
"""
import math
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import channel as ch
from . import shared_funcs as sfun
from .engine import DPEngine, dp_epilogue, dp_epilogue_compact  # noqa: F401


@dataclass
class DPRun:
    """One sweep point (the per-run arguments of processing())."""
    SNR: float
    nu: float
    theta_diff: float
    theta: float
    lr_optim: float
    symb_rate: float
    seed: int = None


_POOL = None


def host_threads():
    """Host threads THIS process may use for the host-side channel model: VAEQ_CPU_THREADS if set, else the affinity mask capped by
    the cgroup CPU quota, divided by the number of ranks sharing the node (LOCAL_WORLD_SIZE, else WORLD_SIZE): under
    torch.distributed.run with 8 ranks each rank takes an eighth of the cores instead of all of them."""
    if os.environ.get("VAEQ_CPU_THREADS"):
        return max(1, int(os.environ["VAEQ_CPU_THREADS"]))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(q) // int(p)))
    except (OSError, ValueError):
        pass
    ranks = int(os.environ.get("LOCAL_WORLD_SIZE") or os.environ.get("WORLD_SIZE") or 1)
    return max(1, min(n // max(1, ranks), 32))


def _host_pool():
    """Thread pool for the host-side channel model (sized to this rank's share of the host cores, see host_threads)."""
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=host_threads())
    return _POOL


def fresh_seed():
    """Entropy for an unseeded run batch on a device generator: distinct per call, per process and per rank."""
    return int(np.random.SeedSequence().entropy & 0xFFFFFFFFFFFF)


def resolve_generator(generator, seeded):
    """``generator=None`` (the default of every entry point): the on-device simulator ("hip") for unseeded runs -- the reference seeds nothing
    (shared_funcs.py:75,84), so there is no random stream to be faithful to and the sweep stays on the GPU --, the reference-faithful host
    simulator ("numpy") for seeded runs, whose frames then equal what the reference draws under the same seed (tests/golden)."""
    if generator is None:
        return "numpy" if seeded else "hip"
    return generator


def check_one_symb_rate(runs, generator):
    """The device generators take one symbol rate per call; a batch that mixes them would silently simulate runs[0]'s."""
    if generator != "numpy" and len({float(r.symb_rate) for r in runs}) > 1:
        raise ValueError(f"generator={generator!r} simulates ONE symb_rate per batch, got {sorted({float(r.symb_rate) for r in runs})}: "
                         "batch the runs by symb_rate (Eval_run_DP.main does) or use generator='numpy'")


_SIDE = {}


def _side_streams(device):
    """The two side streams (channel model, epilogue) of the overlapped frame pipeline, made once per device."""
    key = str(device)
    if key not in _SIDE:
        _SIDE[key] = (torch.cuda.Stream(device), torch.cuda.Stream(device))
    return _SIDE[key]


def _resident_runs(batch_len, sps, M_est, n_lev, threads):
    """How many runs of this shape the device keeps co-resident (vaeq_dp_resident_runs); 0 if the library cannot say."""
    from . import _native as nat
    return max(0, int(nat.lib().vaeq_dp_resident_runs(int(batch_len), int(sps), int(M_est), int(n_lev), int(threads))))


def default_device():
    if not torch.cuda.is_available():
        from ._native import VaeqError
        raise VaeqError("no GPU visible: the VAE training path runs only on the HIP device (there is no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def run_dp_batch(runs, mod, sps, M_est, batch_len, N_frame_max, num_frames, flex_step, channel, tau_cd, tau_pmd, phiIQ,
                 N_lrhalf, flex=False, device=None, generator=None, verbose=False, threads=0, keep_last=False):
    """Train + evaluate R runs.  Returns dict(SER[R,4,num_frames], Var_est[R,2,num_frames], var[R,2]) on the CPU.

    generator: None    = "hip" when no run carries a seed, "numpy" otherwise (resolve_generator);
               "numpy" = reference-faithful host simulator per run (seeded per run when DPRun.seed is set);
               "hip"   = on-device simulator, HIP kernels + hipFFT, Philox streams keyed by the first run's seed (row f1);
               "torch" = batched on-device simulator (channel.generate_batch_gpu), seeded from the first run's seed.
    """
    device = default_device() if device is None else torch.device(device)
    R = len(runs)
    generator = resolve_generator(generator, any(r.seed is not None for r in runs))
    check_one_symb_rate(runs, generator)
    tab_of = {nu: sfun.qam_tables(mod, nu) for nu in {r.nu for r in runs}}          # a sweep has a handful of shaping factors, not one per run
    tabs = [tab_of[r.nu] for r in runs]
    h_channel = sfun.upsampled_channel(channel, sps)
    amps = tabs[0]["amps"]
    amp = torch.tensor(amps, dtype=torch.float32, device=device)
    n_lev = amp.numel()
    P = np.stack([t["P"] for t in tabs])
    nu_sc = np.array([t["nu_sc"] for t in tabs])
    pow_mean = np.array([t["pow_mean"] for t in tabs])
    var_np = np.stack([np.full(2, t["pow_mean"] / 10 ** (r.SNR / 10) / 2) for t, r in zip(tabs, runs)]).astype(np.float32)
    var = torch.tensor(var_np, device=device)
    nu_sc_t = torch.tensor(nu_sc, dtype=torch.float32, device=device)
    eng = DPEngine(R, M_est, amp, P, var, nu_sc, device, sps, threads)

    # frame geometry: func_VAELE_DP_MQAM_shaping.py:38-39 / func_VAEflex_DP_MQAM_shaping.py:37-40
    m_max = N_frame_max // batch_len
    N_frame = m_max * batch_len
    if flex:
        N_out = (N_frame - batch_len) // flex_step * flex_step
        steps, stride, k0, klen = N_out // flex_step, flex_step, (batch_len - flex_step) // 2, flex_step
    else:
        N_out, steps, stride, k0, klen = N_frame, m_max, batch_len, 0, batch_len

    theta = np.array([r.theta for r in runs], dtype=np.float64)
    theta_diff = np.array([r.theta_diff for r in runs], dtype=np.float64)
    lr0 = np.array([r.lr_optim for r in runs], dtype=np.float32)
    lr0_t, lr0_half_t = torch.tensor(lr0, device=device), torch.tensor(lr0 * 0.5, device=device)
    streams = [ch.SeededStreams(r.seed) if r.seed is not None else None for r in runs]
    tgen = None
    if generator == "torch":
        tgen = torch.Generator(device=device)
        tgen.manual_seed(int(runs[0].seed) if runs[0].seed is not None else torch.seed())

    # per-frame results stay on the device until the end: no host synchronisation inside the frame loop (unless verbose)
    SER = torch.empty(R, 4, num_frames, dtype=torch.float32, device=device)
    Var_est = torch.empty(R, 2, num_frames, dtype=torch.float32, device=device)
    last = None
    SNRs = np.array([r.SNR for r in runs], dtype=np.float32)
    hip_seed = (int(runs[0].seed) if runs[0].seed is not None else fresh_seed()) if generator == "hip" else None
    state = {"theta": theta}

    def make_frame(frame):
        """rx[R,2,2,S], data[R,2,2,N] of one frame (channel model of frame `frame`; advances the runs' polarisation angle, :50-51)."""
        th = state["theta"]
        if generator == "hip":                                                  # HIP generator kernels (row f1)
            rx, data = ch.generate_batch_hip(R, N_frame, amps, P, SNRs, h_channel, runs[0].symb_rate, sps, tau_cd, tau_pmd, phiIQ,
                                             th, device, hip_seed, frame)
        elif generator == "torch":
            rx, data = ch.generate_batch_gpu(R, N_frame, amps, P, SNRs, h_channel, runs[0].symb_rate, sps, tau_cd, tau_pmd, phiIQ, th,
                                             device, generator=tgen)
        else:
            def host_frame(i):                                  # the reference-faithful numpy channel model, one run
                r, st = runs[i], streams[i]
                return ch.generate_data_shaping(N_frame, amps, r.SNR, h_channel, tabs[i]["P"], 2, r.symb_rate, sps, tau_cd, tau_pmd, phiIQ,
                                                th[i], "cpu", rng=st.next_rng() if st else None, noise=st.noise if st else None)[:2]
            # seeded runs own their random streams, so they can be generated concurrently (numpy releases the GIL in the FFTs and
            # convolutions); unseeded runs share numpy's global stream like the reference and stay sequential
            if R > 1 and all(st is not None for st in streams):
                pairs = list(_host_pool().map(host_frame, range(R)))
            else:
                pairs = [host_frame(i) for i in range(R)]
            rx = torch.stack([p[0] for p in pairs]).to(device, non_blocking=True)
            data = torch.stack([p[1] for p in pairs]).to(device, non_blocking=True)
        state["theta"] = th + theta_diff                                        # :51
        if flex:
            data = data[:, :, :, batch_len // 2:N_out + batch_len // 2]          # func_VAEflex...:51
        return rx, data

    def train_frame(frame, rx, need_q):
        # lr schedule: group 0 (W) only, set (not multiplied) to lr/2 (func_VAELE_DP_MQAM_shaping.py:45-46)
        # -> lr from frame 0, lr/2 from frame N_lrhalf on (every later trigger re-sets the same value)
        cur_lr_W = lr0_half_t if frame >= N_lrhalf else lr0_t                 # device tensors made once: no H2D copy (a host sync) per frame
        # q itself is only materialised when the caller wants it back (keep_last): the epilogue reads E_q[x_I] and argmax(q), which
        # the training kernel writes directly (5 instead of 32 floats per polarisation symbol through HBM)
        return eng.train(rx, batch_len, steps, cur_lr_W, lr0_t, stride=stride, keep_off=k0, keep_len=klen, want_q=need_q, want_compact=True)

    def finish_frame(frame, out, data):
        ve = out["var_est"][:, 0]                                               # [R,2,steps]
        Var_est[:, :, frame] = ve.mean(dim=2)                                   # :69
        res = dp_epilogue_compact(out["eq"][:, 0], out["dec"][:, 0], out["y"][:, 0], data, amp, nu_sc_t, var, None if flex else batch_len)
        SER[:, :, frame] = res["SER"]
        return res, ve

    # Small batches (fewer runs than the device keeps resident: the script-faithful sweeps, 15 ... 300 runs) leave most of the chip idle
    # while the training kernel walks its 100 dependent steps per frame: there the three stages of a frame run on three streams -- the
    # channel model of frame f + 1 and the epilogue of frame f - 1 beside the training launch of frame f.  Same kernels on the same data
    # in the same per-stream order: results are bit-identical to the serial order (tests/test_processing_gpu.py).  At saturating batch
    # sizes the trainer owns every SIMD's register file and the stages can only time-slice (measured): serial order.
    overlap = (generator == "hip" and not verbose and not keep_last and num_frames > 1 and not os.environ.get("VAEQ_SERIAL_FRAMES")
               and R < max(1, int(_resident_runs(batch_len, sps, M_est, n_lev, threads))))
    if overlap:
        main = torch.cuda.current_stream(device)
        s_gen, s_epi = _side_streams(device)
        s_gen.wait_stream(main)
        s_epi.wait_stream(main)                                                 # SER / Var_est / the tables above were made on `main`

        def gen_async(frame):
            with torch.cuda.stream(s_gen):
                rx, data = make_frame(frame)
                ev = torch.cuda.Event()
                ev.record(s_gen)
            rx.record_stream(main)                                              # allocated on s_gen, consumed on main / s_epi: the caching
            data.record_stream(s_epi)                                           # allocator must not hand the blocks out before those are done
            return rx, data, ev
        nxt = gen_async(0)
        for frame in range(num_frames):
            rx, data, ev = nxt
            if frame + 1 < num_frames:
                nxt = gen_async(frame + 1)
            main.wait_event(ev)
            out = train_frame(frame, rx, False)
            ev_t = torch.cuda.Event()
            ev_t.record(main)
            for k in ("eq", "dec", "y", "var_est"):
                out[k].record_stream(s_epi)
            with torch.cuda.stream(s_epi):
                s_epi.wait_event(ev_t)
                finish_frame(frame, out, data)
        main.wait_stream(s_epi)
        main.wait_stream(s_gen)
    else:
        for frame in range(num_frames):
            rx, data = make_frame(frame)
            need_q = keep_last and frame == num_frames - 1
            out = train_frame(frame, rx, need_q)
            res, ve = finish_frame(frame, out, data)
            if verbose:
                loss = out["loss"][:, 0, -1].cpu()
                snr_est = torch.tensor(pow_mean, dtype=torch.float32) / ve.mean(dim=(1, 2)).cpu()   # :68
                for i in range(R):
                    tag = f"[run {i}] " if R > 1 else ""
                    print(f"{tag}{frame}", "\t\ttraining: loss = ", loss[i].item(), "\tshift_x = ", res["shift_c"][i, 0].item(),
                          "\tshift_y = ", res["shift_c"][i, 1].item(), "\tr = ", int(res["r_c"][i]), "\tSNR_est = ",
                          10 * math.log10(snr_est[i].item()))
                    print("\t\t\t\t\t\t\tSER_x = ", SER[i, 0, frame].item(), "\tSER_y = ", SER[i, 1, frame].item(), "\t(constell. with shaping)")
                    print("\t\t\t\t\t\t\tSER_x = ", SER[i, 2, frame].item(), "\tSER_y = ", SER[i, 3, frame].item(), "\t(soft demapper)")
            if keep_last and frame == num_frames - 1:
                last = dict(q=out["q"][:, 0] if need_q else None, y=out["y"][:, 0], data=data, rx=rx, **res)
    ret = dict(SER=SER.cpu(), Var_est=Var_est.cpu(), var=torch.tensor(var_np), engine=eng)
    if last is not None:
        ret["last"] = last
    return ret
