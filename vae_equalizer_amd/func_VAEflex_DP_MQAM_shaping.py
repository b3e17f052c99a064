"""Drop-in for optical_DP_channel/func_VAEflex_DP_MQAM_shaping.py (:16,90): the same butterfly FIR trained on
overlapping windows of ``batch_len`` symbols advancing by ``flex_step``; only the centre ``flex_step`` outputs of each
window are kept (:59-65)."""
from .dp_runs import DPRun, run_dp_batch


def processing(mod, sps, SNR, nu, M_est, theta_diff, theta, lr_optim, batch_len, N_train_max, num_frames, flex_step, channel,
               symb_rate, tau_cd, tau_pmd, phiIQ, N_lrhalf, *, seed=None, device=None, verbose=True, generator=None):
    """One DP VAEflex run -> (SER_valid[4,num_frames], Var_est[2,num_frames], var[2]), CPU float32 tensors."""
    r = run_dp_batch([DPRun(SNR, nu, theta_diff, theta, lr_optim, symb_rate, seed)], mod, sps, M_est, batch_len, N_train_max,
                     num_frames, flex_step, channel, tau_cd, tau_pmd, phiIQ, N_lrhalf, flex=True, device=device,
                     generator=generator, verbose=verbose)
    if verbose:
        print("We are using the following device for learning:", r["engine"].device)
    return r["SER"][0], r["Var_est"][0], r["var"][0]
