"""Drop-in for optical_DP_channel/func_VAELE_DP_MQAM_shaping.py: same ``processing`` signature and return triple
(:17,95), computed by the fused HIP kernel.  Extra keyword-only arguments never change the positional contract."""
from .dp_runs import DPRun, run_dp_batch


def processing(mod, sps, SNR, nu, M_est, theta_diff, theta, lr_optim, batch_len, N_frame_max, num_frames, flex_step, channel,
               symb_rate, tau_cd, tau_pmd, phiIQ, N_lrhalf, *, seed=None, device=None, verbose=True, generator=None):
    """One DP VAE-LE Monte-Carlo run -> (SER_valid[4,num_frames], Var_est[2,num_frames], var[2]), CPU float32 tensors.

    Rows of SER_valid: 0-1 constellation-based SER x/y, 2-3 soft-demapper SER x/y (:79,89).
    seed: None = unseeded like the reference; int = reproducible frames (channel.SeededStreams)."""
    r = run_dp_batch([DPRun(SNR, nu, theta_diff, theta, lr_optim, symb_rate, seed)], mod, sps, M_est, batch_len, N_frame_max,
                     num_frames, flex_step, channel, tau_cd, tau_pmd, phiIQ, N_lrhalf, flex=False, device=device,
                     generator=generator, verbose=verbose)
    if verbose:
        print("We are using the following device for learning:", r["engine"].device)
    return r["SER"][0], r["Var_est"][0], r["var"][0]
