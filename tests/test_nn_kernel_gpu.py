"""GPU parity of the AWGN VAE-NN kernels (vaeq_nn_train / vaeq_nn_forward) against the reference's golden vectors
(G8, captured from AWGN_channel/func_VAENN_MQAM.py) and the CPU oracle."""
import numpy as np
import pytest
import torch

import oracle
from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu
G8 = ["G8_vaenn_64qam", "G8_vaenn_16qam_small", "G8_vaenn_4qam_k5"]
DEV = "cuda:0"


def _np(t):
    return t.detach().cpu().numpy()


def _engine(g, R=1):
    from vae_equalizer_amd.engine import NNEngine
    eng = NNEngine(R, int(g["M_est"]), int(g["k1"]), int(g["k2"]), g["amp_levels"], DEV, int(g["sps"]))
    assert eng.NP == g["theta0"].size
    eng.theta.copy_(torch.from_numpy(g["theta0"]).to(DEV).expand(R, -1))
    return eng


@pytest.mark.parametrize("name", G8)
def test_nn_teacher_forced_step(name):
    """q, ELBO, the gradient of all parameters (fc1/fc2 weights + biases, h_est) and the AMSGrad update on the captured minibatch."""
    g = load_golden(name)
    B, sps = int(g["B"]), int(g["sps"])
    eng = _engine(g)
    rx = torch.from_numpy(g["rx"][None, :, :B * sps]).to(DEV)
    r = eng.train(rx, B, 1, float(g["lr"]), want_q=True, debug_grads=True)
    torch.cuda.synchronize()
    assert np.max(np.abs(_np(r["q"])[0] - g["q0"])) < 5e-6
    assert abs(_np(r["loss"])[0, 0] - g["loss"][0]) / abs(g["loss"][0]) < 1e-5
    o = eng.offsets()
    t = oracle.nn_step_grads(g["rx"][:, :B * sps], g["theta0"], g["amp_levels"], int(g["k1"]), int(g["k2"]), int(g["M_est"]), sps, np.float64)
    for a, b in zip(o[:-1], o[1:]):                                       # per parameter tensor
        assert relerr(_np(r["g"])[0, a:b], g["g0"][a:b]) < 2e-4
        assert relerr(_np(r["g"])[0, a:b], t["g"][a:b]) < max(3 * relerr(g["g0"][a:b], t["g"][a:b]), 5e-6)
    lr = float(g["lr"])
    ok = np.abs(g["g0"]) > 1e-4 * np.abs(g["g0"]).max()                    # rounding-level gradients move by +-lr on a coin flip
    assert np.max(np.abs(_np(eng.theta)[0] - g["theta1"])[ok]) < 5e-6
    assert np.max(np.abs(_np(eng.theta)[0] - g["theta1"])) < 2.01 * lr
    assert int(eng.step[0]) == 1


@pytest.mark.parametrize("name", G8)
def test_nn_free_run(name):
    g = load_golden(name)
    B, ns, lr = int(g["B"]), int(g["n_steps"]), float(g["lr"])
    eng = _engine(g, R=3)                                                  # three identical runs in one launch
    r = eng.train(torch.from_numpy(g["rx"][None]).to(DEV).expand(3, -1, -1).contiguous(), B, ns, lr)
    torch.cuda.synchronize()
    loss = _np(r["loss"])
    assert np.array_equal(loss[0], loss[1]) and np.array_equal(loss[0], loss[2])     # deterministic, run-independent
    assert np.max(np.abs(loss[0, :3] - g["loss"][:3]) / np.abs(g["loss"][:3])) < 2e-5
    assert np.max(np.abs(loss[0] - g["loss"]) / np.abs(g["loss"])) < 2e-3
    assert np.max(np.abs(_np(eng.theta)[0] - g[f"theta{ns}"])) < (1e-4 if ns <= 3 else 2 * ns * lr)
    st = oracle.NNState(g["theta0"], np.float32)
    lo = oracle.nn_train(st, g["rx"], ns, B, g["amp_levels"], int(g["k1"]), int(g["k2"]), int(g["M_est"]), lr, int(g["sps"]), np.float32)
    assert np.max(np.abs(loss[0, :3] - lo[:3]) / np.abs(lo[:3])) < 2e-5
    assert relerr(_np(eng.vmax)[0], g["vmax"]) < 5e-3 and int(eng.step[0]) == ns


@pytest.mark.parametrize("name,N", [("G8_vaenn_64qam", 1500), ("G8_vaenn_16qam_small", 180), ("G8_vaenn_4qam_k5", 123)])
def test_nn_forward_tiles_match_oracle(name, N):
    """Eval-mode forward over a long block, computed in 256-symbol tiles with real neighbours in the halos, == the oracle's
    single-pass forward (zero padding only at the block's ends)."""
    g = load_golden(name)
    sps = int(g["sps"])
    eng = _engine(g)
    eng.theta.copy_(torch.from_numpy(g["theta3"])[None])
    x = g["rx"][:, :N * sps]
    q = eng.forward(torch.from_numpy(x[None]).to(DEV))
    qo = oracle.nn_forward(x, g["theta3"], len(g["amp_levels"]), int(g["k1"]), int(g["k2"]), sps, np.float64)
    assert np.max(np.abs(_np(q)[0] - qo)) < 5e-6
