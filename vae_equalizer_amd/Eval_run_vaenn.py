"""Drop-in for AWGN_channel/Eval_run_vaenn.py: same constants, sweep order, result tensor and ``.mat`` schema (:36, :58-68); sweep
points are sharded over ranks (r mod world) and gathered once at the end."""
from datetime import datetime
from itertools import product

import scipy.io as io
import torch

mod = '64-QAM'          # Modulation Format: {4,16,64}-QAM
sps = 2                 # samples per symbol
net_type_vec = ['Net']  # ['Net_BN'] # topology
channel = 'h1'          # 'h2'
M_vec = [25]            # taps of the estimated channel impulse response
k1_vec, k2_vec = [25], [3]  # kernel size (of layer1, layer2)
batch_len_vec = [300]   # length of training/updating batch in symbols
lr_optim_vec = [4e-3]
SNR_vec = [24]
iter = 3                # independent runs per setting
N_valid = 15000         # symbols per evaluation step
train_len = 4000        # training symbols per epoch
num_epochs = 500
epe = 2                 # epochs per evaluation

savePATH = ""
base_seed = None        # int -> reproducible runs; None = like the reference
generator = "hip"       # "hip": on-device channel model; "numpy": host restatement per run


def sweep_points():
    """The reference's loop nest (:38-56) for one net_type."""
    for (n, batch_len), (l, lr), (m, M), (a, k1), (b, k2), (s, SNR), i in product(enumerate(batch_len_vec), enumerate(lr_optim_vec),
                                                                                  enumerate(M_vec), enumerate(k1_vec), enumerate(k2_vec),
                                                                                  enumerate(SNR_vec), range(iter)):
        yield (s, b, a, m, l, n, i), dict(batch_len=batch_len, lr=lr, M=M, k1=k1, k2=k2, SNR=SNR)


def main():
    from . import sweep
    from .func_VAENN_MQAM import run_vaenn_batch

    rank, world, local_rank = sweep.init_distributed()
    device = sweep.device_for_rank(local_rank, world)
    if rank == 0:
        print('Run code on: ', device, f'({world} rank(s))')
    name = save_dict = None
    for net_type in net_type_vec:
        points = list(sweep_points())
        mine = sweep.my_slice(len(points), rank, world)
        local = torch.zeros(len(mine), num_epochs // epe, dtype=torch.float32)
        key = lambda p: (p["M"], p["k1"], p["k2"], p["batch_len"])
        for b, shape in enumerate(sorted({key(points[i][1]) for i in mine})):       # one batch per problem shape
            sel = [k for k, i in enumerate(mine) if key(points[i][1]) == shape]
            runs = [dict(SNR=points[mine[k]][1]["SNR"], lr_optim=points[mine[k]][1]["lr"],
                         seed=None if base_seed is None else base_seed + 1000 * mine[k]) for k in sel]
            M, k1, k2, batch_len = shape
            local[sel] = run_vaenn_batch(runs, mod, sps, M, k1, k2, batch_len, N_valid, train_len, num_epochs, epe, channel, device=device,
                                         generator=generator, seed=sweep.stream_seed(base_seed, rank, b + 1000 * net_type_vec.index(net_type)), net_type=net_type)
        rows = sweep.gather_rows(local, len(points), rank, world)
        if rank != 0:
            continue
        SER = torch.empty(len(SNR_vec), len(k2_vec), len(k1_vec), len(M_vec), len(lr_optim_vec), len(batch_len_vec), iter, num_epochs // epe,
                          dtype=torch.float32)
        for k, (idx, _) in enumerate(points):
            SER[idx] = rows[k]
        name = f"{savePATH}SERvsSNR_{net_type}_{channel}_{mod}_{sps}_{N_valid}_{epe}_{train_len}_{datetime.today().strftime('%y%m%d%H%M%S')}.mat"
        save_dict = {'SER': SER.numpy(), 'SNR': SNR_vec, 'k2': k2_vec, 'k1': k1_vec, 'M': M_vec, 'lr': lr_optim_vec, 'N_train': batch_len_vec}
        io.savemat(name, {'dict': save_dict})
    return (name, save_dict) if rank == 0 else None


if __name__ == "__main__":
    main()
