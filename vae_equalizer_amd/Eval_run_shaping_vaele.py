"""Drop-in for AWGN_channel/Eval_run_shaping_vaele.py: same constants, sweep order, result tensor and ``.mat`` schema
(:40, :60-69); sweep points are sharded over ranks (r mod world) and gathered once at the end."""
from datetime import datetime
from itertools import product

import scipy.io as io
import torch

mod = '64-QAM'          # Modulation Format: {4,16,64}-QAM
sps = 2                 # samples per symbol
channel = 'h1'          # 'h2'
M_vec = [25]            # taps of the estimated channel impulse response
N_train_vec = [350]     # length of training/updating batch in symbols
lr_optim_vec = [5e-3]
SNR_vec = [24]
nu_vec = [0]            # [0] [0.0270955] [0.0872449] [0.1222578]
iter = 20               # independent runs per setting
N_valid = 15000         # symbols per evaluation step
train_len = 1200        # training symbols per epoch
num_epochs = 500
epe = 2                 # epochs per evaluation

savePATH = ""
base_seed = None        # int -> reproducible runs; None = like the reference
generator = None        # None: "hip" (on-device generator vaeq_gen_awgn) for unseeded sweeps, "numpy" (reference-faithful host simulator) when base_seed is set


def sweep_points():
    """The reference's loop nest (:43-58); note its SER has no nu axis, so later nu values overwrite earlier ones."""
    for (n, N_train), (l, lr), (m, M), (s, SNR), nu, i in product(enumerate(N_train_vec), enumerate(lr_optim_vec), enumerate(M_vec),
                                                                 enumerate(SNR_vec), nu_vec, range(iter)):
        yield (s, 0, 0, m, l, n, i), dict(N_train=N_train, lr=lr, M=M, SNR=SNR, nu=nu)


def main():
    from . import sweep
    from .func_VAELE_MQAM_shaping import run_awgn_batch

    rank, world, local_rank = sweep.init_distributed()
    device = sweep.device_for_rank(local_rank, world)
    if rank == 0:
        print('Run code on: ', device, f'({world} rank(s))')
    points = list(sweep_points())
    mine = sweep.my_slice(len(points), rank, world)
    local = torch.zeros(len(mine), num_epochs // epe, dtype=torch.float32)
    for b, (M, N_train) in enumerate(sorted({(points[i][1]["M"], points[i][1]["N_train"]) for i in mine})):   # one batch per problem shape
        sel = [k for k, i in enumerate(mine) if (points[i][1]["M"], points[i][1]["N_train"]) == (M, N_train)]
        runs = [dict(SNR=points[mine[k]][1]["SNR"], nu=points[mine[k]][1]["nu"], lr_optim=points[mine[k]][1]["lr"],
                     seed=None if base_seed is None else base_seed + 1000 * mine[k]) for k in sel]
        local[sel] = run_awgn_batch(runs, mod, sps, M, N_train, N_valid, train_len, num_epochs, epe, channel, device=device,
                                    generator=generator, seed=sweep.stream_seed(base_seed, rank, b))
    rows = sweep.gather_rows(local, len(points), rank, world)
    if rank != 0:
        return None
    SER = torch.empty(len(SNR_vec), 1, 1, len(M_vec), len(lr_optim_vec), len(N_train_vec), iter, num_epochs // epe, dtype=torch.float32)
    for k, (idx, _) in enumerate(points):
        SER[idx] = rows[k]
    nu = nu_vec[-1]
    name = f"{savePATH}SERvsSNR_VAELE_shaping_{nu}_{channel}_{mod}_{sps}_{N_valid}_{epe}_{train_len}_{datetime.today().strftime('%y%m%d%H%M%S')}.mat"
    save_dict = {'SER': SER.numpy(), 'SNR': SNR_vec, 'M': M_vec, 'lr': lr_optim_vec, 'N_train': N_train_vec, 'nu': nu_vec}
    io.savemat(name, {'dict': save_dict})
    return name, save_dict


if __name__ == "__main__":
    main()
