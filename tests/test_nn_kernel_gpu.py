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


@pytest.mark.parametrize("B,sps,k1,k2,M,bn", [(64, 2, 5, 1, 9, False), (100, 1, 9, 5, 13, False), (40, 2, 25, 9, 9, False), (90, 3, 7, 3, 11, False),
                                              (130, 2, 63, 3, 25, False), (34, 2, 3, 3, 5, True), (150, 2, 11, 5, 25, True)])
def test_nn_64qam_mfma_shapes_match_oracle(B, sps, k1, k2, M, bn):
    """64-QAM (16 channels): the MFMA path of the two convolutions, their weight gradients and the transposed convolution on ragged
    shapes -- one teacher-forced step (q, loss, every gradient) against the fp64 oracle, Net and Net_BN."""
    from vae_equalizer_amd.engine import NNEngine
    rng = np.random.default_rng(B + k1)
    n = 8
    lev = np.arange(-(n - 1), n, 2).astype(np.float32)
    amp = (lev / np.sqrt(np.mean(lev ** 2) * 2)).astype(np.float32)
    eng = NNEngine(1, M, k1, k2, amp, DEV, sps, batch_norm=bn)
    eng.init_parameters()
    theta0 = (_np(eng.theta)[0] + 0.02 * rng.standard_normal(eng.NP)).astype(np.float32)
    eng.theta.copy_(torch.from_numpy(theta0)[None].to(DEV))
    x = (0.5 * rng.standard_normal((2, B * sps))).astype(np.float32)
    r = eng.train(torch.from_numpy(x[None]).to(DEV), B, 1, 1e-3, want_q=True, debug_grads=True, no_update=True)
    torch.cuda.synchronize()
    if bn:
        t = oracle.nnbn_step_grads(x, theta0, np.concatenate([np.zeros(2 * n), np.ones(2 * n)]), amp, k1, k2, M, sps, np.float64)
    else:
        t = oracle.nn_step_grads(x, theta0, amp, k1, k2, M, sps, np.float64)
    assert np.max(np.abs(_np(r["q"])[0] - t["q"])) < 2e-5
    assert abs(_np(r["loss"])[0, 0] - t["loss"]) / abs(t["loss"]) < 1e-5
    o = eng.offsets()
    for a, b in zip(o[:-1], o[1:]):
        assert relerr(_np(r["g"])[0, a:b], t["g"][a:b]) < 1e-4, (a, b)


@pytest.mark.parametrize("N,sps,k1,k2", [(700, 2, 7, 5), (255, 2, 25, 3), (256, 1, 9, 1), (511, 3, 5, 3), (1000, 2, 63, 9)])
def test_nn_64qam_eval_forward_shapes_match_oracle(N, sps, k1, k2):
    """64-QAM eval forward (MFMA convolutions, 255-symbol tiles with real neighbours in the halos) on ragged shapes == the oracle's
    single-pass forward."""
    from vae_equalizer_amd.engine import NNEngine
    rng = np.random.default_rng(N + k1)
    n, M = 8, 9
    lev = np.arange(-(n - 1), n, 2).astype(np.float32)
    amp = (lev / np.sqrt(np.mean(lev ** 2) * 2)).astype(np.float32)
    eng = NNEngine(1, M, k1, k2, amp, DEV, sps)
    eng.init_parameters()
    theta = (_np(eng.theta)[0] + 0.02 * rng.standard_normal(eng.NP)).astype(np.float32)
    eng.theta.copy_(torch.from_numpy(theta)[None].to(DEV))
    x = (0.5 * rng.standard_normal((2, N * sps))).astype(np.float32)
    q = eng.forward(torch.from_numpy(x[None]).to(DEV))
    qo = oracle.nn_forward(x, theta, n, k1, k2, sps, np.float64)
    assert np.max(np.abs(_np(q)[0] - qo)) < 2e-5


@pytest.mark.parametrize("name,N", [("G8_vaenn_64qam", 1500), ("G8_vaenn_16qam_small", 180), ("G8_vaenn_4qam_k5", 123)])
def test_nn_forward_tiles_match_oracle(name, N):
    """Eval-mode forward over a long block, computed in 255-symbol tiles with real neighbours in the halos, == the oracle's
    single-pass forward (zero padding only at the block's ends)."""
    g = load_golden(name)
    sps = int(g["sps"])
    eng = _engine(g)
    eng.theta.copy_(torch.from_numpy(g["theta3"])[None])
    x = g["rx"][:, :N * sps]
    q = eng.forward(torch.from_numpy(x[None]).to(DEV))
    qo = oracle.nn_forward(x, g["theta3"], len(g["amp_levels"]), int(g["k1"]), int(g["k2"]), sps, np.float64)
    assert np.max(np.abs(_np(q)[0] - qo)) < 5e-6


@pytest.mark.parametrize("name,N", [("G8_vaenn_64qam", 3000), ("G8_vaenn_16qam_small", 180)])
def test_nn_validate_matches_torch_mirror(name, N):
    """forward + find_shift + SER_q fused (vaeq_nn_validate) == the torch restatement of the reference's three calls on the q of
    vaeq_nn_forward; runs differ in the delay of the TX reference and in a quadrant rotation of the decisions."""
    from vae_equalizer_amd.func_VAELE_MQAM_shaping import SER_q, find_shift
    g = load_golden(name)
    sps, n = int(g["sps"]), len(g["amp_levels"])
    R = 4
    eng = _engine(g, R)
    eng.theta.copy_(torch.from_numpy(g[f"theta{int(g['n_steps'])}"])[None].expand(R, -1))
    x = torch.from_numpy(g["rx"][:, :N * sps])[None].expand(R, -1, -1).contiguous().to(DEV)
    rng = np.random.default_rng(3)
    q = eng.forward(x)
    # a TX reference the decisions agree with up to 2 % errors, delayed / rotated per run: data = levels of argmax(q) rolled by k
    dec = torch.stack([q[0, :n].argmax(0), q[0, n:].argmax(0)]).cpu().numpy()
    flip = rng.random(dec.shape) < 0.02
    dec = np.where(flip, (dec + 1) % n, dec)
    amps = g["amp_levels"]
    datas = []
    for r, (k, rot) in enumerate([(0, 0), (4, 1), (-7, 2), (9, 3)]):
        d = np.roll(dec, -k, axis=1)                                          # q[:, 11+k+j] lines up with data[:, 11+j]
        K = n - 1
        d = [d, np.stack([K - d[0], K - d[1]]), np.stack([d[1], K - d[0]]), np.stack([K - d[1], d[0]])][rot]
        datas.append(amps[d])
    data = torch.from_numpy(np.stack(datas)).to(torch.float16).to(DEV)
    ser, sh = eng.validate(x, data, 21)
    torch.cuda.synchronize()
    amp = torch.tensor(amps, dtype=torch.float32, device=DEV)
    for i in range(R):
        s_ref = int(find_shift(q[i], data[i], 21, amp, n))
        assert int(sh[i]) == s_ref, (i, int(sh[i]), s_ref)
        e_ref = float(SER_q(q[i][:, 11 + s_ref:-11], data[i][:, 11:-11 - s_ref], sps, n))
        assert abs(float(ser[i]) - e_ref) <= 2.0 / N, (i, float(ser[i]), e_ref)
    if N >= 1000:
        assert [int(v) for v in sh] == [0, 4, -7, 9] and float(ser.max()) < 0.06


# ------------------------------------------------------------------ Net_BN (BatchNorm variant), G11
G11 = ["G11_vaennbn_64qam", "G11_vaennbn_16qam_small"]


def _bn_engine(g, R=1):
    from vae_equalizer_amd.engine import NNEngine
    eng = NNEngine(R, int(g["M_est"]), int(g["k1"]), int(g["k2"]), g["amp_levels"], DEV, int(g["sps"]), batch_norm=True)
    assert eng.NP == g["theta0"].size
    eng.theta.copy_(torch.from_numpy(g["theta0"]).to(DEV).expand(R, -1))
    eng.bn.copy_(torch.from_numpy(g["bn0"]).to(DEV).expand(R, -1))
    return eng


@pytest.mark.parametrize("name", G11)
def test_nnbn_teacher_forced_step(name):
    """Net_BN: q with batch statistics, ELBO, the gradient of all parameters incl. BatchNorm's weight and bias, the running
    statistics after the step, the AMSGrad update."""
    g = load_golden(name)
    B, sps = int(g["B"]), int(g["sps"])
    eng = _bn_engine(g)
    rx = torch.from_numpy(g["rx"][None, :, :B * sps]).to(DEV)
    r = eng.train(rx, B, 1, float(g["lr"]), want_q=True, debug_grads=True)
    torch.cuda.synchronize()
    assert np.max(np.abs(_np(r["q"])[0] - g["q0"])) < 1e-5
    assert abs(_np(r["loss"])[0, 0] - g["loss"][0]) / abs(g["loss"][0]) < 1e-5
    assert relerr(_np(eng.bn)[0], g["bn1"]) < 5e-6
    o = eng.offsets()
    assert len(o) == 8
    for a, b in zip(o[:-1], o[1:]):
        assert relerr(_np(r["g"])[0, a:b], g["g0"][a:b]) < 3e-4, (a, b)
    ok = np.abs(g["g0"]) > 1e-4 * np.abs(g["g0"]).max()
    assert np.max(np.abs(_np(eng.theta)[0] - g["theta1"])[ok]) < 1e-5
    assert np.max(np.abs(_np(eng.theta)[0] - g["theta1"])) < 2.01 * float(g["lr"])


@pytest.mark.parametrize("name", G11)
def test_nnbn_free_run_and_eval_forward(name):
    g = load_golden(name)
    B, ns, lr, sps = int(g["B"]), int(g["n_steps"]), float(g["lr"]), int(g["sps"])
    eng = _bn_engine(g, R=2)
    r = eng.train(torch.from_numpy(g["rx"][None]).to(DEV).expand(2, -1, -1).contiguous(), B, ns, lr)
    torch.cuda.synchronize()
    loss = _np(r["loss"])
    assert np.array_equal(loss[0], loss[1])
    assert np.max(np.abs(loss[0, :2] - g["loss"][:2]) / np.abs(g["loss"][:2])) < 2e-5
    assert np.max(np.abs(loss[0] - g["loss"]) / np.abs(g["loss"])) < 2e-3
    assert relerr(_np(eng.bn)[0], g[f"bn{ns}"]) < 1e-3
    assert np.max(np.abs(_np(eng.theta)[0] - g[f"theta{ns}"])) < 2 * ns * lr
    # eval mode (net.eval(): running statistics) with the reference's own parameters and statistics
    eng.theta.copy_(torch.from_numpy(g[f"theta{ns}"])[None].expand(2, -1))
    eng.bn.copy_(torch.from_numpy(g[f"bn{ns}"])[None].expand(2, -1))
    Ne = B * min(ns, 3)
    q = eng.forward(torch.from_numpy(g["rx"][None, :, :Ne * sps]).to(DEV).expand(2, -1, -1).contiguous())
    assert np.max(np.abs(_np(q)[0] - g["q_eval"])) < 1e-5


def test_nn_half_minibatch_kernel_matches_whole_minibatch_kernel(monkeypatch):
    """VAEQ_NN_HALF=1 runs 64-QAM `Net` at the sweep script's shape on nn_train_half_kernel (fc1 activations of half a minibatch in LDS, one
    recomputation, AMSGrad vectors streamed from the caller's arrays, two workgroups per CU) instead of nn_train_kernel (everything of a minibatch in
    LDS; the default: the half-minibatch form measured no faster).  Same math, different summation order of the weight gradients (half by half): teacher-forced gradients agree to rounding, five free steps
    of 40 independently initialised runs to 2e-5 in the parameters, the AMSGrad vectors and the step counts likewise; both are deterministic."""
    from vae_equalizer_amd.engine import NNEngine
    from vae_equalizer_amd.func_VAENN_MQAM import vaenn_tables
    t = vaenn_tables("64-QAM", "h1", 2)
    R, B, steps = 40, 300, 5
    gen = torch.Generator(device=DEV); gen.manual_seed(5)
    rx = 0.5 * torch.randn(R, 2, steps * B * 2, device=DEV, generator=gen)

    def run(half, no_update=False, n=steps):
        monkeypatch.setenv("VAEQ_NN_HALF", "1" if half else "0")
        eng = NNEngine(R, 25, 25, 3, t["amps"], DEV, 2)
        g2 = torch.Generator(device=DEV); g2.manual_seed(11)
        eng.init_parameters(g2)
        r = eng.train(rx, B, n, 4e-3, want_q=True, debug_grads=True, no_update=no_update)
        torch.cuda.synchronize()
        return eng, r

    _, a = run(True, no_update=True, n=1)
    _, b = run(False, no_update=True, n=1)
    assert torch.equal(a["q"], b["q"])                                                 # the forward pass is the same arithmetic;
    assert np.max(np.abs(_np(a["loss"]) - _np(b["loss"])) / np.abs(_np(b["loss"]))) < 1e-6    # the loss sums meet in a different number of waves
    ga, gb = _np(a["g"]), _np(b["g"])
    assert np.max(np.abs(ga - gb)) < 2e-6 * np.max(np.abs(gb))
    ea, a = run(True)
    eb, b = run(False)
    assert np.max(np.abs(_np(a["loss"]) - _np(b["loss"])) / np.abs(_np(b["loss"]))) < 2e-5
    assert np.max(np.abs(_np(ea.theta) - _np(eb.theta))) < 2e-5
    for x, y in ((ea.m, eb.m), (ea.v, eb.v), (ea.vmax, eb.vmax)):
        assert np.max(np.abs(_np(x) - _np(y))) < 1e-5 * max(1.0, float(y.abs().max()))
    assert torch.equal(ea.step, eb.step) and int(ea.step[0]) == steps
    ec, c = run(True)
    assert torch.equal(a["loss"], c["loss"]) and torch.equal(ea.theta, ec.theta) and torch.equal(ea.vmax, ec.vmax)   # bitwise reproducible


@pytest.mark.parametrize("B,sps,k1,k2,M", [(64, 2, 5, 1, 9), (100, 1, 9, 5, 13), (40, 2, 25, 9, 9), (90, 3, 7, 3, 11), (130, 2, 63, 3, 25), (250, 2, 25, 3, 25), (17, 2, 3, 1, 5)])
def test_nn_64qam_free_steps_on_ragged_shapes_match_oracle(B, sps, k1, k2, M):
    """64-QAM `Net`, four FREE steps on ragged shapes against the fp32 oracle: after every update the owner of a convolution weight writes it into the
    transposed copies the MFMA convolutions read (walk order of the forward pass, [k][c][cc] copy of the backward pass) -- a wrong index there shows from
    the second step on.  Losses to 2e-5, parameters to 2e-5, AMSGrad vectors to 1e-4 of their maximum."""
    from vae_equalizer_amd.engine import NNEngine
    rng = np.random.default_rng(7 * B + k1)
    n, steps, lr = 8, 4, 2e-3
    lev = np.arange(-(n - 1), n, 2).astype(np.float32)
    amp = (lev / np.sqrt(np.mean(lev ** 2) * 2)).astype(np.float32)
    eng = NNEngine(2, M, k1, k2, amp, DEV, sps)
    eng.init_parameters()
    theta0 = (_np(eng.theta) + 0.02 * rng.standard_normal((2, eng.NP))).astype(np.float32)
    eng.theta.copy_(torch.from_numpy(theta0).to(DEV))
    x = (0.5 * rng.standard_normal((2, 2, steps * B * sps))).astype(np.float32)
    r = eng.train(torch.from_numpy(x).to(DEV), B, steps, lr)
    torch.cuda.synchronize()
    for i in range(2):
        st = oracle.NNState(theta0[i], np.float32)
        lo = oracle.nn_train(st, x[i], steps, B, amp, k1, k2, M, lr, sps, np.float32)
        assert np.max(np.abs(_np(r["loss"])[i] - lo) / np.abs(lo)) < 2e-5
        assert np.max(np.abs(_np(eng.theta)[i] - st.theta)) < 2e-5
        for ours, ref in ((eng.m, st.m), (eng.v, st.v), (eng.vmax, st.vmax)):
            assert np.max(np.abs(_np(ours)[i] - ref)) < 1e-4 * max(np.max(np.abs(ref)), 1e-12)
    assert int(eng.step[0]) == steps
