"""the constant-modulus baselines of the DP scripts (func_CMA_DP_MQAM_shaping.py, func_CMAbatch_DP_MQAM_shaping.py,
func_CMAflex_DP_MQAM_shaping.py), R independent runs at once.

Per frame: channel model -> vaeq_cma (HIP: CMA / CMAbatch / CMAflex) -> cut 10 symbols at both ends -> vaeq_cpe (HIP, Viterbi-Viterbi)
-> the reference's two-stage epilogue (find_shift_symb_full + SER_constell_shaping on the phase-corrected output, then soft_dec on
the ALIGNED output whose kept window SER_constell_shaping has normalised in place, find_shift and SER_IQflip;
func_CMA_DP_MQAM_shaping.py:39-53) with the batched torch restatements of
epilogue.py and the HIP soft demapper."""
import math

import numpy as np
import torch

from . import channel as ch
from . import epilogue as epi
from . import shared_funcs as sfun
from .dp_runs import DPRun, _host_pool, check_one_symb_rate, default_device, fresh_seed, resolve_generator  # noqa: F401
from .engine import cma, cpe, soft_demap

N_CUT = 10   # symbols cut at both frame ends before the phase estimation (func_CMA_DP_MQAM_shaping.py:26,39)


def cma_frame_epilogue(out_const, data, amp, nu_sc, var):
    """out_const[R,2,2,K] (CMA output of one frame), data[R,2,2,K] fp16 -> dict(SER[R,4], shift_c, r_c, shift_q, r_q): phase estimation
    (vaeq_cpe) + the two-stage epilogue in one HIP launch (vaeq_cma_epilogue; q is never materialised)."""
    from .engine import cma_epilogue
    y = cpe(out_const[..., N_CUT:-N_CUT].contiguous())                          # :39
    return cma_epilogue(y, data[..., N_CUT:-N_CUT], amp, nu_sc, var)            # :40-52


def cma_frame_epilogue_torch(out_const, data, amp, nu_sc, var):
    """The same from the batched torch restatements of the reference's functions + the HIP soft demapper (materialises q): the form the fused
    kernel is checked against; also returns y = the aligned output with its kept window normalised (what the reference leaves in out_const)."""
    y = cpe(out_const[..., N_CUT:-N_CUT].contiguous())                          # :39
    d = data[..., N_CUT:-N_CUT]                                                 # :40
    R, N = y.shape[0], y.shape[-1]
    shift_c, r_c = epi.shift_search(y[:, :, 0], d)                              # :41
    ya = epi._align(y, shift_c, r_c)                                            # :42-43
    mask_c = epi._keep_mask(shift_c, N, None, y.device)                         # [11 : -11 - max|shift|], :44
    ser_c, scale_n = epi.ser_constellation(ya, d, mask_c, amp, nu_sc, var[:, 0], return_scale=True)
    # :44 passes a slice VIEW of out_const, and SER_constell_shaping normalises its argument in place (shared_funcs.py:242): from here on
    # the kept window of out_const carries the mean-radius normalisation, the 11 + max|shift| edge symbols keep the raw CMA scale
    ya = torch.where(mask_c.reshape(R, 1, 1, N), ya * scale_n.reshape(R, 1, 1, 1), ya)
    q = soft_demap(ya.contiguous(), amp, var, nu_sc)                            # :48, on the aligned, window-normalised output
    n = amp.numel()
    Eq = torch.einsum("i,rpin->rpn", amp, q[:, :, :n])
    shift_q, r_q = epi.shift_search(Eq, d)                                      # :49
    dec = torch.stack([q[:, :, :n].argmax(dim=2), q[:, :, n:].argmax(dim=2)], dim=2)
    dec = epi._align(dec, shift_q, r_q)                                         # :50-51
    ser_q = epi.ser_soft_demap(dec, d, epi._keep_mask(shift_q, N, None, y.device), n)   # :52
    return dict(SER=torch.cat([ser_c, ser_q], dim=1), shift_c=shift_c, r_c=r_c, shift_q=shift_q, r_q=r_q, y=ya)


def run_cma_batch(runs, mode, mod, sps, M_est, batch_len, N_train_max, num_frames, flex_step, channel, tau_cd, tau_pmd, phiIQ, N_lrhalf,
                  device=None, generator=None, verbose=False):
    """R baseline runs (list of dp_runs.DPRun; lr_optim = the CMA step size) -> dict(SER[R,4,num_frames], Var_est[R,2,num_frames] (zeros,
    like the reference), var[R,2], h).  mode: "CMA" | "CMAbatch" | "CMAflex"."""
    if mode not in ("CMA", "CMAbatch", "CMAflex"):
        raise ValueError(f"unknown CMA variant {mode!r}")
    device = default_device() if device is None else torch.device(device)
    R = len(runs)
    generator = resolve_generator(generator, any(r.seed is not None for r in runs))   # unseeded -> on-device simulator
    tabs = [sfun.qam_tables(mod, r.nu) for r in runs]
    h_channel = sfun.upsampled_channel(channel, sps)
    amps = tabs[0]["amps"]
    amp = torch.tensor(amps, dtype=torch.float32, device=device)
    var_np = np.stack([np.full(2, t["pow_mean"] / 10 ** (r.SNR / 10) / 2) for t, r in zip(tabs, runs)]).astype(np.float32)
    var = torch.tensor(var_np, device=device)
    nu_sc = torch.tensor([t["nu_sc"] for t in tabs], dtype=torch.float32, device=device)
    h = torch.zeros(R, 2, 2, 2, M_est, dtype=torch.float32, device=device)     # sfun.init: Dirac on the straight paths
    h[:, 0, 0, 0, M_est // 2] = 1
    h[:, 1, 1, 0, M_est // 2] = 1
    theta = np.array([r.theta for r in runs], dtype=np.float64)
    theta_diff = np.array([r.theta_diff for r in runs], dtype=np.float64)
    lr = np.array([r.lr_optim for r in runs], dtype=np.float64)
    streams = [ch.SeededStreams(r.seed) if r.seed is not None else None for r in runs]
    SER = torch.empty(R, 4, num_frames, dtype=torch.float32, device=device)
    P = np.stack([t["P"] for t in tabs])
    check_one_symb_rate(runs, generator)
    hip_seed = int(runs[0].seed) if R and runs[0].seed is not None else fresh_seed()   # unseeded: fresh entropy per call and rank
    for frame in range(num_frames):
        if frame % N_lrhalf == 0 and frame != 0:                                # :30-31
            lr = lr * 0.5
        if generator == "hip":
            rx, data = ch.generate_batch_hip(R, N_train_max, amps, P, np.array([r.SNR for r in runs], np.float32), h_channel, runs[0].symb_rate,
                                             sps, tau_cd, tau_pmd, phiIQ, theta, device, hip_seed, frame)
        elif generator == "numpy":
            def host_frame(i):
                r, st = runs[i], streams[i]
                return ch.generate_data_shaping(N_train_max, amps, r.SNR, h_channel, tabs[i]["P"], 2, r.symb_rate, sps, tau_cd, tau_pmd, phiIQ,
                                                theta[i], "cpu", rng=st.next_rng() if st else None, noise=st.noise if st else None)[:2]
            seeded = R > 1 and all(st is not None for st in streams)
            pairs = list(_host_pool().map(host_frame, range(R))) if seeded else [host_frame(i) for i in range(R)]
            rx = torch.stack([p[0] for p in pairs]).to(device)
            data = torch.stack([p[1] for p in pairs]).to(device)
        else:
            raise ValueError(f"unknown generator {generator!r}")
        theta = theta + theta_diff
        out_const, e = cma(rx, h, lr.astype(np.float32), sps, mode, batch_len, flex_step, 1.0, want_e=verbose)
        res = cma_frame_epilogue(out_const, data, amp, nu_sc, var)
        SER[:, :, frame] = res["SER"]
        if verbose:
            es, ser_h = e.sum(dim=(1, 2)).cpu(), res["SER"].cpu()
            for i in range(R):
                tag = f"[run {i}] " if R > 1 else ""
                print(f"{tag}{frame}", "\t\ttraining: loss = ", es[i].item(), "\tshift_x = ", res["shift_c"][i, 0].item(), "\tshift_y = ",
                      res["shift_c"][i, 1].item(), "\tr = ", int(res["r_c"][i]))
                print("\t\t\t\t\t\t\tSER_x = ", ser_h[i, 0].item(), "\tSER_y = ", ser_h[i, 1].item(), "\t(constell. with shaping)")
                print("\t\t\t\t\t\t\tSER_x = ", ser_h[i, 2].item(), "\tSER_y = ", ser_h[i, 3].item(), "\t(soft demapper)")
    return dict(SER=SER.cpu(), Var_est=torch.zeros(R, 2, num_frames), var=torch.tensor(var_np), h=h)


def _processing(mode, mod, sps, SNR, nu, M_est, theta_diff, theta, lr_optim, batch_len, N_train_max, num_frames, flex_step, channel, symb_rate,
                tau_cd, tau_pmd, phiIQ, N_lrhalf, seed, device, verbose, generator):
    r = run_cma_batch([DPRun(SNR, nu, theta_diff, theta, lr_optim, symb_rate, seed)], mode, mod, sps, M_est, batch_len, N_train_max, num_frames,
                      flex_step, channel, tau_cd, tau_pmd, phiIQ, N_lrhalf, device=device, generator=generator, verbose=verbose)
    return r["SER"][0], r["Var_est"][0], r["var"][0]
