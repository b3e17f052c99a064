"""Drop-in for optical_DP_channel/func_CMA_DP_MQAM_shaping.py: same ``processing`` signature (:17) and return value
(:58): (SER_valid[4,num_frames], Var_est[2,num_frames] (zeros), var[2]); the CMA loop and the phase estimation run as HIP kernels
(cma_runs.run_cma_batch: vaeq_cma, vaeq_cpe)."""
from .cma_runs import _processing


def processing(mod, sps, SNR, nu, M_est, theta_diff, theta, lr_optim, batch_len, N_train_max, num_frames, flex_step, channel, symb_rate, tau_cd,
               tau_pmd, phiIQ, N_lrhalf, *, seed=None, device=None, verbose=True, generator=None):
    return _processing("CMA", mod, sps, SNR, nu, M_est, theta_diff, theta, lr_optim, batch_len, N_train_max, num_frames, flex_step, channel,
                       symb_rate, tau_cd, tau_pmd, phiIQ, N_lrhalf, seed, device, verbose, generator)
