"""The reference's operator-level training loop, written exactly like func_VAELE_DP_MQAM_shaping.py:57-66 against the mirrors
(twoXtwoFIR module, loss_function_shaping, torch.optim.Adam, loss.backward()), runs on HIP kernels and reproduces the reference's
gradients and taps (G1 / G2 fixtures)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G1 = ["G1_dp_step_64qam_pcs", "G1_dp_step_64qam", "G1_dp_step_16qam", "G1_dp_step_4qam", "G1_dp_step_64qam_nu0872", "G1_dp_step_64qam_nu1222"]   # the last two: config 5's heavy shaping (Eval_run_DP.py:24)


@pytest.mark.parametrize("name", G1)
def test_operator_level_backward_matches_reference_autograd(name):
    from vae_equalizer_amd import shared_funcs as sfun
    g = load_golden(name)
    B, sps, M = int(g["B"]), int(g["sps"]), int(g["M_est"])
    net = sfun.twoXtwoFIR(M, sps).to(DEV)
    with torch.no_grad():
        net.conv_w.weight.copy_(torch.from_numpy(g["W0"]))
    h_est = torch.tensor(g["h0"], device=DEV, requires_grad=True)
    amp, P, var = (torch.from_numpy(g[k]).to(DEV) for k in ("amp_levels", "P", "var"))
    opt = torch.optim.Adam(net.parameters(), lr=float(g["lr"]))
    opt.add_param_group({"params": h_est})
    rx = torch.from_numpy(g["rx"]).to(DEV)
    for s in range(3):
        mb = rx[:, :, s * B * sps:(s + 1) * B * sps].clone()
        opt.zero_grad()
        q, out = net(mb, amp, var, float(g["nu_sc"]))
        loss, var_est = sfun.loss_function_shaping(q.squeeze(), mb.squeeze(), h_est, amp, P)
        loss.backward()
        if s == 0:
            assert abs(loss.item() - g["loss0"]) / abs(g["loss0"]) < 1e-5
            assert relerr(var_est.cpu().numpy(), g["var_est0"]) < 1e-5 and not var_est.requires_grad
            assert relerr(h_est.grad.cpu().numpy(), g["gh0"]) < 2e-5
            assert relerr(net.conv_w.weight.grad.cpu().numpy(), g["gW0"]) < 1e-4
        opt.step()
    assert np.max(np.abs(net.conv_w.weight.detach().cpu().numpy() - g["W3"])) < 2e-5
    assert np.max(np.abs(h_est.detach().cpu().numpy() - g["h3"])) < 2e-5


def test_operator_level_loop_equals_fused_kernel():
    """20 free steps from the Dirac start through the operator-level loop == the fused kernel == the reference (G2)."""
    from vae_equalizer_amd import shared_funcs as sfun
    g = load_golden("G2_dp_freerun")
    B, sps, M = int(g["B"]), int(g["sps"]), int(g["M_est"])
    net = sfun.twoXtwoFIR(M, sps).to(DEV)
    h_est, h_ch, P, amp_levels, amps, pol, nu_sc, var, pow_mean = sfun.init("h0", "64-QAM", DEV, 0.0, sps, M, 23)
    opt = torch.optim.Adam(net.parameters(), lr=float(g["lr"]))
    opt.add_param_group({"params": h_est})
    P_t = torch.tensor(P, dtype=torch.float32, device=DEV)
    rx = torch.from_numpy(g["rx"]).to(DEV)
    losses = []
    for m in range(20):
        mb = rx[:, :, m * B * sps:(m + 1) * B * sps]
        opt.zero_grad()
        q, out = net(mb, amp_levels, var, nu_sc)
        loss, _ = sfun.loss_function_shaping(q, mb, h_est, amp_levels, P_t)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert np.max(np.abs(np.array(losses) - g["loss"][:20]) / np.abs(g["loss"][:20])) < 1e-5
    assert np.max(np.abs(net.conv_w.weight.detach().cpu().numpy() - g["W_after20"])) < 1e-5
    assert np.max(np.abs(h_est.detach().cpu().numpy() - g["h_after20"])) < 1e-5


def test_eval_mode_builds_no_graph():
    from vae_equalizer_amd import shared_funcs as sfun
    g = load_golden(G1[0])
    net = sfun.twoXtwoFIR(int(g["M_est"]), 2).to(DEV)
    with torch.no_grad():
        q, out = net(torch.from_numpy(g["rx"][:, :, :200]).to(DEV), torch.from_numpy(g["amp_levels"]).to(DEV), torch.from_numpy(g["var"]).to(DEV), 0.0)
    assert not q.requires_grad and not out.requires_grad


# ------------------------------------------------------------------ the AWGN pair (twoFIR / loss_function) and the VAE-NN loss
@pytest.mark.parametrize("name", ["G4_awgn_16qam_cfg1", "G4_awgn_64qam_pcs_free10", "G4_awgn_4qam_small"])
def test_awgn_operator_level_loop_matches_reference(name):
    """The reference's AWGN loop (func_VAELE_MQAM_shaping.py:297-306) written against the mirrors: twoFIR -> loss_function -> backward ->
    Adam(amsgrad).step with HIP forward/backward kernels reproduces the reference's gradients and first update."""
    from vae_equalizer_amd import func_VAELE_MQAM_shaping as aw
    g = load_golden(name)
    B, sps, M = int(g["B"]), int(g["sps"]), int(g["M_est"])
    x = torch.from_numpy(g["rx"][:, :B * sps]).to(DEV)
    amp, P = torch.from_numpy(g["amp_levels"]).to(DEV), torch.from_numpy(g["P"]).to(DEV)
    net = aw.twoFIR(M, sps).to(DEV)
    with torch.no_grad():
        net.conv_w.weight.copy_(torch.from_numpy(g["W0"]).reshape(1, 2, M))
    h = torch.tensor(g["h0"], device=DEV, requires_grad=True)
    opt = torch.optim.Adam(net.parameters(), lr=float(g["lr"]), amsgrad=True)
    opt.add_param_group({"params": h})
    opt.zero_grad()
    q, out = net(x, amp, float(g["amp_mean"]), float(g["var"]))
    loss = aw.loss_function(q, x, h, DEV, amp, P)
    loss.backward()
    assert abs(float(loss.detach()) - g["loss"][0]) / abs(g["loss"][0]) < 1e-5
    assert relerr(net.conv_w.weight.grad.cpu().numpy().reshape(2, -1), g["gW0"].reshape(2, -1)) < 2e-4
    assert relerr(h.grad.cpu().numpy(), g["gh0"]) < 2e-5
    opt.step()
    ok = np.abs(g["gW0"].reshape(2, -1)) > 1e-6 * np.abs(g["gW0"]).max()
    assert np.max(np.abs(net.conv_w.weight.detach().cpu().numpy().reshape(2, -1) - g["W1"].reshape(2, -1))[ok]) < 1e-5
    assert np.max(np.abs(h.detach().cpu().numpy() - g["h1"])) < 1e-5


def test_vaenn_reference_loop_with_torch_net_and_hip_loss():
    """func_VAENN_MQAM's loop with the reference's own kind of torch Net (Conv1d / ELU / softmax on the GPU) and the HIP loss_function
    with autograd: the gradient of every parameter equals the reference's (G8)."""
    import torch.nn as nn
    import torch.nn.functional as F
    from vae_equalizer_amd import func_VAENN_MQAM as nn_
    g = load_golden("G8_vaenn_16qam_small")
    B, sps, k1, k2, M = int(g["B"]), int(g["sps"]), int(g["k1"]), int(g["k2"]), int(g["M_est"])
    n = len(g["amp_levels"])
    C_ = 2 * n
    o = np.cumsum([0, C_ * 2 * k1, C_, C_ * C_ * k2, C_, 2 * M])
    th = torch.from_numpy(g["theta0"]).to(DEV)
    fc1 = nn.Conv1d(2, C_, k1, padding=k1 // 2).to(DEV)
    fc2 = nn.Conv1d(C_, C_, k2, padding=k2 // 2, stride=sps).to(DEV)
    with torch.no_grad():
        fc1.weight.copy_(th[o[0]:o[1]].reshape(C_, 2, k1)); fc1.bias.copy_(th[o[1]:o[2]])
        fc2.weight.copy_(th[o[2]:o[3]].reshape(C_, C_, k2)); fc2.bias.copy_(th[o[3]:o[4]])
    h = th[o[4]:o[5]].reshape(2, M).clone().requires_grad_(True)
    x = torch.from_numpy(g["rx"][:, :B * sps]).to(DEV)
    a2 = fc2(F.elu(fc1(x[None])))[0]
    q = torch.cat([torch.softmax(a2[:n], 0), torch.softmax(a2[n:], 0)])
    loss = nn_.loss_function(q, x, h, DEV, torch.from_numpy(g["amp_levels"]).to(DEV))
    loss.backward()
    assert abs(float(loss.detach()) - g["loss"][0]) / abs(g["loss"][0]) < 1e-5
    got = torch.cat([fc1.weight.grad.reshape(-1), fc1.bias.grad, fc2.weight.grad.reshape(-1), fc2.bias.grad, h.grad.reshape(-1)]).cpu().numpy()
    for a, b in zip(o[:-1], o[1:]):
        assert relerr(got[a:b], g["g0"][a:b]) < 5e-4, (a, b)
