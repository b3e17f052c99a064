"""Drop-in for optical_DP_channel/Eval_run_DP.py: the same constants, the same nested sweep, the same result tensors and
``.mat`` schema (:52-54, :99-114) -- but the sweep points are flattened into a batch, sharded over the node's GPUs
(``python -m torch.distributed.run --nproc-per-node 8 -m vae_equalizer_amd.Eval_run_DP``) and trained by the fused HIP
kernel, one launch per frame for all runs of a rank.  Edit the constants below exactly like in the reference.
"""
from datetime import datetime
from itertools import product

import numpy as np
import scipy.io as io
import torch

mod = '64-QAM'  # modulation format:  {4,16,64}-QAM
sps = 2         # oversampling factor in samples per symbol

loss_type = 'VAE'  # 'VAE' 'VAEflex' 'CMA' 'CMAbatch' 'CMAflex'
channel = 'h0'     # optical channel with PMD and ISI caused by CD

nu_vec = [0]  # [0] [0.0270955] [0.0872449] [0.1222578]: PCS entropies 6, 5.72, 4.6, 4.125 bit (PCS-64-QAM)

symb_rate_vec = [90e9]  # symbol rate in Baud

tau_pmd = 0.1e-12 * np.sqrt(1000)  # PMD coefficient
tau_cd = -26e-24 * 1               # residual chromatic dispersion
phiIQ = np.array([0.0314, 0.0314], dtype=np.complex64)  # static IQ-shift in rad
theta_vec = [np.pi / 10]           # HV shift in rad
theta_diff_vec = [0.06 * np.pi]    # HV shift drift per frame

SNR_vec = [23]  # np.arange(20,30,2)

M_vec = [25]            # filter length (taps)
batch_len_vec = [100]   # minibatch length in symbols
flex_step_vec = [10]    # window step of the flex scheme in symbols
lr_optim_vec = [2.5e-3, 2e-3, 3e-3]

iter = 5            # independent runs per setting
N_lrhalf = 170      # frames until the learning rate is halved
num_frames = 170
N_frame_max = 10000

savePATH = ""
base_seed = None    # int -> reproducible runs (run i of the flattened sweep uses base_seed + 1000*i); None = like the reference
generator = None    # None: "hip" (on-device channel simulator) for unseeded sweeps, "numpy" (reference-faithful host simulator) when base_seed is set; or force "numpy" / "hip" / "torch"


def sweep_points():
    """The reference's loop nest (:68-86) flattened: yields (index tuple into SER[...], run arguments)."""
    for (n, nu), (nt, batch_len), (l, lr), (m, M), (t1, td), (sr, rate), (ss, fs), (v, th), (s, SNR), i in product(
            enumerate(nu_vec), enumerate(batch_len_vec), enumerate(lr_optim_vec), enumerate(M_vec), enumerate(theta_diff_vec),
            enumerate(symb_rate_vec), enumerate(flex_step_vec), enumerate(theta_vec), enumerate(SNR_vec), range(iter)):
        yield (s, sr, n, t1, m, l, nt, ss, v, i), dict(SNR=SNR, nu=nu, theta_diff=td, theta=th, lr_optim=lr, symb_rate=rate,
                                                         M=M, batch_len=batch_len, flex_step=fs)


def main():
    from .dp_runs import DPRun, run_dp_batch
    from . import sweep

    rank, world, local_rank = sweep.init_distributed()
    device = sweep.device_for_rank(local_rank, world)
    if rank == 0:
        print('Run code on: ', device, f'({world} rank(s))')
    shape_tail = (len(SNR_vec), len(symb_rate_vec), len(nu_vec), len(theta_diff_vec), len(M_vec), len(lr_optim_vec),
                  len(batch_len_vec), len(flex_step_vec), len(theta_vec), iter)
    SER = torch.empty(4, *shape_tail, num_frames, dtype=torch.float32)
    Var_est = torch.empty(2, *shape_tail, num_frames, dtype=torch.float32)
    var_real = torch.empty(2, *shape_tail, 1, dtype=torch.float32)

    points = list(sweep_points())
    rows = torch.zeros(len(points), 8, num_frames, dtype=torch.float32)   # per run: SER[4] | Var_est[2] | var[2] (broadcast)
    mine = sweep.my_slice(len(points), rank, world)
    # one batch per problem shape (M, batch_len, flex_step) and symbol rate (the device generators simulate one rate per call)
    key = lambda i: (points[i][1]["M"], points[i][1]["batch_len"], points[i][1]["flex_step"], points[i][1]["symb_rate"])
    shapes = sorted({key(i) for i in mine})
    local = torch.zeros(len(mine), 8, num_frames, dtype=torch.float32)
    for (M, batch_len, fs, rate) in shapes:
        sel = [k for k, i in enumerate(mine) if key(i) == (M, batch_len, fs, rate)]
        runs = [DPRun(points[mine[k]][1]["SNR"], points[mine[k]][1]["nu"], points[mine[k]][1]["theta_diff"], points[mine[k]][1]["theta"],
                      points[mine[k]][1]["lr_optim"], points[mine[k]][1]["symb_rate"],
                      None if base_seed is None else base_seed + 1000 * mine[k]) for k in sel]
        if loss_type in ('CMA', 'CMAbatch', 'CMAflex'):         # the constant-modulus baselines (:58-65)
            from .cma_runs import run_cma_batch
            r = run_cma_batch(runs, loss_type, mod, sps, M, batch_len, N_frame_max, num_frames, fs, channel, tau_cd, tau_pmd, phiIQ, N_lrhalf,
                              device=device, generator=generator if generator in (None, "numpy", "hip") else "hip", verbose=False)
        elif loss_type in ('VAE', 'VAEflex'):
            r = run_dp_batch(runs, mod, sps, M, batch_len, N_frame_max, num_frames, fs, channel, tau_cd, tau_pmd, phiIQ, N_lrhalf,
                             flex=(loss_type == 'VAEflex'), device=device, generator=generator, verbose=False)
        else:
            raise NameError(f"loss_type {loss_type!r}: the reference leaves `process` undefined (:56-65)")
        local[sel, 0:4] = r["SER"]
        local[sel, 4:6] = r["Var_est"]
        local[sel, 6:8] = r["var"].unsqueeze(-1).expand(-1, -1, num_frames)
    rows = sweep.gather_rows(local, len(points), rank, world)
    if rank != 0:
        return None
    for k, (idx, _) in enumerate(points):
        SER[(slice(None),) + idx] = rows[k, 0:4]
        Var_est[(slice(None),) + idx] = rows[k, 4:6]
        var_real[(slice(None),) + idx + (0,)] = rows[k, 6:8, 0]
    name = f"{savePATH}SERvsSNR_{loss_type}_DP_{mod}_N_lrhalf_{N_lrhalf}_N_train_{N_frame_max}_{datetime.today().strftime('%y%m%d%H%M%S')}.mat"
    save_dict = {'SER': SER.numpy(), 'Var_est': Var_est.numpy(), 'var_real': var_real.numpy(), 'SNR': SNR_vec, 'nu': nu_vec,
                 'theta_diff': theta_diff_vec, 'theta': theta_vec, 'M': M_vec, 'lr': lr_optim_vec, 'batch_len': batch_len_vec,
                 'symb_rate': symb_rate_vec, 'symb_step': flex_step_vec}
    io.savemat(name, {'dict': save_dict})
    return name, save_dict


if __name__ == "__main__":
    main()
