"""Edge cases of the C ABI on the GPU: empty batches, NULL / inconsistent arguments, the largest supported tap count, long
minibatches, sps != 2 -- each compute case against the CPU oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle
from conftest import relerr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
OK, ERR_NULL, ERR_SHAPE, ERR_LDS = 0, -1, -2, -3


def _amp(n):
    a = np.arange(-(n - 1), n, 2).astype(np.float32)
    return a / np.sqrt(2 * np.mean(a ** 2))


def test_empty_batches_are_a_no_op():
    """R = 0 is legal everywhere (a rank may own no sweep point): VAEQ_OK, nothing launched, nothing touched."""
    from vae_equalizer_amd import _native as nat
    from vae_equalizer_amd.engine import AWGNEngine, DPEngine, dp_epilogue_compact
    L = nat.lib()
    eng = DPEngine(0, 25, _amp(8), np.zeros((0, 8), np.float32), np.zeros((0, 2), np.float32), np.zeros(0, np.float32), DEV)
    out = eng.train(torch.zeros(0, 1, 2, 2, 400, device=DEV), 100, 2, 1e-3, want_compact=True)
    assert out["q"].shape == (0, 1, 2, 16, 200) and out["loss"].shape == (0, 1, 2)
    res = dp_epilogue_compact(out["eq"][:, 0], out["dec"][:, 0], out["y"][:, 0], torch.zeros(0, 2, 2, 200, dtype=torch.float16, device=DEV),
                              _amp(8), 0.0, 0.01, 100)
    assert res["SER"].shape == (0, 4)
    aeng = AWGNEngine(0, 25, _amp(8), np.zeros((0, 8), np.float32), [], [], DEV)
    assert aeng.train(torch.zeros(0, 2, 700, device=DEV), 350, 1, 1e-3)["loss"].shape == (0, 1)
    ser, sh, _ = aeng.validate(torch.zeros(0, 2, 2000, device=DEV), torch.zeros(0, 2, 1000, dtype=torch.float16, device=DEV))
    assert ser.shape == (0,) and sh.shape == (0,)
    z = C.c_void_p(torch.zeros(8, device=DEV).data_ptr())
    assert L.vaeq_gen_awgn(0, 100, 141, 2, 8, 40, 242, 12, z, z, z, z, C.c_uint64(1), C.c_uint32(0), z, z, z, None, None, None, None) == OK


def test_null_and_inconsistent_arguments_are_refused():
    from vae_equalizer_amd import _native as nat
    L = nat.lib()
    t = torch.zeros(4096, device=DEV)
    p = C.c_void_p(t.data_ptr())
    i32 = C.c_void_p(torch.zeros(4, dtype=torch.int32, device=DEV).data_ptr())
    base = dict(R=1, n_frames=1, steps=1, B=100, sps=2, M=25, n_lev=8, stride_sym=100, keep_off=0, keep_len=100, S=400, rx=p, W=p, h=p,
                adam_mW=p, adam_vW=p, adam_mh=p, adam_vh=p, step=i32, amp=p, P=p, var=p, nu_sc=p, lr_W=p, lr_h=p, threads=0, no_update=1)
    call = lambda **kw: L.vaeq_dp_train(C.byref(nat.DPArgs(**{**base, **kw})), None)
    assert call(rx=None) == ERR_NULL and call(W=None) == ERR_NULL and call(step=None) == ERR_NULL
    assert call(M=24) == ERR_SHAPE                                  # even tap count
    assert call(n_lev=3) == ERR_SHAPE and call(n_lev=16) == ERR_SHAPE
    assert call(S=199) == ERR_SHAPE                                 # window past the row
    assert call(B=24) == ERR_SHAPE                                  # B <= 2 * (M // 2): empty ELBO
    assert call(keep_off=50, keep_len=60) == ERR_SHAPE              # kept slice past the minibatch
    assert call(threads=7) == ERR_SHAPE
    assert call(B=20000, S=80000, stride_sym=20000, keep_len=20000) == ERR_LDS
    assert L.vaeq_dp_train(None, None) == ERR_NULL
    assert L.vaeq_strerror(ERR_LDS).decode() and L.vaeq_strerror(-99).decode()
    assert L.vaeq_dp_epilogue_compact(1, 1000, 8, 100, None, p, p, p, p, p, p, p, i32, i32, None) == ERR_NULL
    assert L.vaeq_dp_epilogue_compact(1, 30, 8, 0, p, p, p, p, p, p, p, p, i32, i32, None) == ERR_SHAPE        # shorter than the trims
    assert L.vaeq_dp_epilogue_compact(1, 1000, 8, 300, p, p, p, p, p, p, p, p, i32, i32, None) == ERR_SHAPE    # N % batch_len
    assert L.vaeq_awgn_validate(1, 1000, 2, 24, 8, 21, p, p, p, p, p, p, p, p, i32, None) == ERR_SHAPE
    assert L.vaeq_awgn_validate(1, 1000, 2, 25, 8, 64, p, p, p, p, p, p, p, p, i32, None) == ERR_SHAPE         # n_shift > 32
    assert L.vaeq_gen_awgn(1, 100, 141, 2, 8, 40, 999, 12, p, p, p, p, C.c_uint64(1), C.c_uint32(0), p, p, p, None, None, None, None) == ERR_SHAPE
    assert L.vaeq_gen_dp_frame(1, 100, 141, 2, 8, 200, 242, 256, 12, p, p, p, p, p, 1.8e11, 0.0, 0.0, 1.0, 0.0, 1.0, 0.0, C.c_uint64(1),
                               C.c_uint32(0), p, p, p, None, None, None) == ERR_SHAPE                              # Lg > 96 taps


@pytest.mark.parametrize("M,B,sps,n", [(63, 150, 2, 8), (31, 600, 2, 4), (9, 41, 3, 2), (5, 33, 1, 8), (25, 20, 2, 8), (25, 14, 2, 4), (9, 10, 2, 8),
                                       (25, 99, 2, 8), (25, 21, 2, 8), (13, 155, 2, 4), (31, 333, 2, 2), (25, 777, 2, 8)])
def test_dp_extreme_shapes_against_oracle(M, B, sps, n):
    """Largest tap count (M = 63), a 600-symbol minibatch (42 KB of LDS), sps = 3 and sps = 1: the generic kernel; minibatches SHORTER than the filter
    (the reference's commented-out batch_len options, Eval_run_DP.py:38: B = 20 with M = 25 -- an empty KL slice, 16 residual samples) and ODD minibatch
    lengths at one to eight wavefronts per run: the wave kernel; 3 free steps."""
    from vae_equalizer_amd.engine import DPEngine
    rng = np.random.default_rng(M * B)
    R, steps = 2, 3
    amp = _amp(n)
    P = np.full((R, n), 1 / n, np.float32)
    var = np.full((R, 2), 0.01, np.float32)
    rx = (0.5 * rng.standard_normal((R, 2, 2, steps * B * sps))).astype(np.float32)
    W0 = (0.03 * rng.standard_normal((R, 2, 4, M))).astype(np.float32)
    W0[:, 0, 0, M // 2] += 1; W0[:, 1, 1, M // 2] += 1
    h0 = (0.03 * rng.standard_normal((R, 2, 2, 2, M))).astype(np.float32)
    h0[:, 0, 0, 0, M // 2] += 1; h0[:, 1, 1, 0, M // 2] += 1
    eng = DPEngine(R, M, amp, P, var, 0.01, DEV, sps)
    eng.set_state(W0, h0)
    r = eng.train(torch.from_numpy(rx).to(DEV), B, steps, 1e-3)
    torch.cuda.synchronize()
    for i in range(R):
        st = oracle.DPState(M, np.float32, W0[i], h0[i])
        o = oracle.dp_train(st, rx[i], steps, B, amp, P[i], var[i], 0.01, 1e-3, 1e-3, sps)
        assert np.max(np.abs(r["loss"][i, 0].cpu().numpy() - o["loss"]) / np.abs(o["loss"])) < 2e-5
        assert relerr(r["y"][i, 0].cpu().numpy(), o["out"]) < 1e-5
        assert np.max(np.abs(r["q"][i, 0].cpu().numpy() - o["q"])) < 1e-4
        assert np.max(np.abs(eng.W[i].cpu().numpy() - st.W)) < 2e-5 and np.max(np.abs(eng.h[i].cpu().numpy() - st.h)) < 2e-5


@pytest.mark.parametrize("M,B,sps,n", [(63, 200, 2, 8), (25, 1000, 2, 4), (9, 41, 3, 2)])
def test_awgn_extreme_shapes_against_oracle(M, B, sps, n):
    from vae_equalizer_amd.engine import AWGNEngine
    rng = np.random.default_rng(M + B)
    steps = 3
    amp = _amp(n)
    P = np.full(n, 1 / n, np.float32)
    amp_mean, var = float(np.mean(np.abs(amp))), 0.02
    rx = (0.5 * rng.standard_normal((2, steps * B * sps))).astype(np.float32)
    W0 = (0.03 * rng.standard_normal((1, 2, M))).astype(np.float32); W0[0, 0, M // 2] += 1
    h0 = (0.03 * rng.standard_normal((2, M))).astype(np.float32); h0[0, M // 2] += 1
    eng = AWGNEngine(1, M, amp, P, amp_mean, var, DEV, sps)
    eng.set_state(W0, h0)
    r = eng.train(torch.from_numpy(rx[None]).to(DEV), B, steps, 1e-3)
    torch.cuda.synchronize()
    st = oracle.AWGNState(M, np.float32, W0, h0)
    loss = oracle.awgn_train(st, rx, steps, B, amp, P, amp_mean, var, 1e-3, sps)
    assert np.max(np.abs(r["loss"][0].cpu().numpy() - loss) / np.abs(loss)) < 2e-5
    assert np.max(np.abs(eng.W[0].cpu().numpy() - st.W[0])) < 2e-5 and np.max(np.abs(eng.h[0].cpu().numpy() - st.h)) < 2e-5


def test_nn_entry_points_reject_bad_arguments_and_accept_empty_batches():
    from vae_equalizer_amd import _native as nat
    from vae_equalizer_amd.engine import NNEngine
    L = nat.lib()
    assert L.vaeq_nn_param_count(25, 8, 25, 3, 0) == 16 * 2 * 25 + 16 + 16 * 16 * 3 + 16 + 50
    assert L.vaeq_nn_param_count(25, 8, 25, 3, 1) == 16 * 2 * 25 + 16 + 16 * 16 * 3 + 16 + 32 + 50                     # + BatchNorm weight, bias
    assert L.vaeq_nn_param_count(24, 8, 25, 3, 0) == ERR_SHAPE and L.vaeq_nn_param_count(25, 8, 24, 3, 0) == ERR_SHAPE
    assert L.vaeq_nn_param_count(25, 3, 25, 3, 0) == ERR_SHAPE and L.vaeq_nn_param_count(25, 8, 25, 11, 0) == ERR_SHAPE
    assert L.vaeq_nn_lds_bytes(300, 2, 25, 8, 25, 3, 0) > 0 and L.vaeq_nn_lds_bytes(20, 2, 25, 8, 25, 3, 0) == ERR_SHAPE  # B <= 2 (M // 2)
    assert L.vaeq_nn_lds_bytes(300, 2, 25, 8, 25, 3, 1) > L.vaeq_nn_lds_bytes(300, 2, 25, 8, 25, 3, 0)
    p = C.c_void_p(torch.zeros(4096, device=DEV).data_ptr())
    i32 = C.c_void_p(torch.zeros(4, dtype=torch.int32, device=DEV).data_ptr())
    base = dict(R=1, steps=1, B=60, sps=2, M=9, n_lev=4, k1=11, k2=3, S=120, rx=p, theta=p, adam_m=p, adam_v=p, adam_x=p, step=i32, amp=p, lr=p,
                no_update=1)
    call = lambda **kw: L.vaeq_nn_train(C.byref(nat.NNArgs(**{**base, **kw})), None)
    assert call() == OK
    assert call(theta=None) == ERR_NULL and call(rx=None) == ERR_NULL
    assert call(S=119) == ERR_SHAPE and call(k1=10) == ERR_SHAPE and call(n_lev=5) == ERR_SHAPE and call(steps=0) == ERR_SHAPE
    assert call(B=3000, S=6000) == ERR_LDS                                       # 16 x 6000 floats of hidden activations do not fit
    assert call(R=0, rx=None, theta=None) == OK
    assert call(batch_norm=1) == ERR_NULL and call(batch_norm=1, bn_running=p) == OK   # Net_BN needs its running statistics
    assert L.vaeq_nn_validate(1, 32, 2, 9, 4, 11, 3, 21, p, p, None, p, p, p, i32, None) == ERR_SHAPE                 # N < 64
    assert L.vaeq_nn_validate(1, 1000, 2, 9, 4, 11, 3, 21, p, None, None, p, p, p, i32, None) == ERR_NULL
    assert L.vaeq_nn_forward(0, 1000, 2, 9, 4, 11, 3, None, None, None, None, None) == OK
    assert L.vaeq_awgn_loss(1, 60, 2, 9, 4, p, p, p, p, None, p, None) == OK and L.vaeq_awgn_loss(1, 8, 2, 9, 4, p, p, p, p, None, p, None) == ERR_SHAPE
    eng = NNEngine(0, 9, 11, 3, _amp(4), DEV)
    assert eng.train(torch.zeros(0, 2, 240, device=DEV), 60, 2, 1e-3)["loss"].shape == (0, 2)
    with pytest.raises(ValueError):
        NNEngine(1, 9, 12, 3, _amp(4), DEV)


def test_engine_checkpoint_resume_is_bit_identical():
    """SURVEY section 5 (checkpoint / resume, absent from the reference): a run's whole carried state is seven tensors; saving them between frames and
    loading them into a fresh engine continues the run bit-identically."""
    from vae_equalizer_amd.engine import DPEngine
    rng = np.random.default_rng(3)
    R, B, M, steps = 5, 100, 25, 6
    amp = (np.arange(-7, 8, 2) / np.sqrt(42.0)).astype(np.float32)
    mk = lambda: DPEngine(R, M, amp, np.full(8, 1 / 8, np.float32), [0.0025, 0.003], 0.0, "cuda:0", 2)
    rx = torch.from_numpy((0.4 * rng.standard_normal((R, 2, 2, 2, steps * B * 2))).astype(np.float32)).cuda()
    a, b = mk(), mk()
    ra = a.train(rx, B, steps, 2.5e-3)                                                  # two frames in one go
    b.train(rx[:, :1].contiguous(), B, steps, 2.5e-3)
    sd = b.state_dict()
    assert all(not t.is_cuda for t in sd.values()) and int(sd["step"][0]) == steps
    c = mk()
    c.load_state_dict(sd)
    rc = c.train(rx[:, 1:].contiguous(), B, steps, 2.5e-3)
    torch.cuda.synchronize()
    assert torch.equal(ra["loss"][:, 1], rc["loss"][:, 0]) and torch.equal(a.W, c.W) and torch.equal(a.h, c.h) and torch.equal(a.vh, c.vh)
    assert torch.equal(a.step, c.step)
    with pytest.raises(ValueError):
        c.load_state_dict({**sd, "W": sd["W"][:2]})
