"""bench.py end to end on the GPU box: the JSON contract (parity gate, both roofline fractions, the kernel name taken from the launch,
extra.pipeline / extra.configs, cpu_baseline) and a two-rank rehearsal of `--gpus 2` (gloo, both ranks on the one GPU of the box)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _last_json(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert lines, stdout[-2000:]
    return json.loads(lines[-1])


def test_bench_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--runs", "512", "--steps", "3", "--warmup", "1", "--min-seconds", "0.3",
                        "--cpu-seconds", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["dtype"] == "f32" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["timed_regions"] >= 2 and d["region_ms"]["min"] <= d["region_ms"]["median"] <= d["region_ms"]["max"]
    assert abs(d["ms_per_step"] * 3 - d["region_ms"]["median"]) < 1e-6 * d["region_ms"]["median"] + 1e-9
    assert abs(d["value"] - 512 * 10000 * 3 / (d["region_ms"]["median"] * 1e-3)) < 1e-6 * d["value"]
    p = d["parity"]
    assert p["ok"] and p["elbo_rel_max"] <= p["tol"] == 1e-5 and p["taps_abs_max"] <= 1e-5 and p["ser_abs_max"] <= 2e-3 and p["runs"] >= 8 and p["steps"] == 10
    rf = d["roofline"]
    assert rf["kernel"] == "vaeq::dp_wave_kernel<25, 8, 100, true, 1, 1, 0>"          # what vaeq_dp_train launched, not a literal in bench.py
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / 8000.0) < 1e-12 and rf["kernel_ms"] <= d["ms_per_step"] * 1.02    # (two medians over different samples)
    assert rf["kernel_ms_minmax"][0] <= rf["kernel_ms"] <= rf["kernel_ms_mean"] * 1.2 and rf["kernel_ms_mean"] <= rf["kernel_ms_minmax"][1]
    assert rf["traffic"] is None and "runs: profiled 8192, this run 512" in rf["traffic_dropped_because"]      # never a stale figure without a reason
    assert abs(rf["achieved"] - 176 * 512 * 10000 / (rf["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * rf["achieved"]
    assert 0 < rf["flop_frac"] < 1 and abs(rf["flop_frac"] - 5043.0 * 512 * 10000 / (rf["kernel_ms"] * 1e-3) / 157.3e12) < 1e-9
    assert 2000 < rf["peak_measured_copy"] < 8000 and rf["frac_of_measured_copy"] > rf["frac"]
    e = d["extra"]
    assert e["pipeline"]["ms_per_frame"] > rf["kernel_ms"] and e["pipeline"]["dp_symbols_per_s"] > 0
    # (wall-clock figures of a 40-frame host loop: the three-stream order is 15-20 % faster in a quiet process (profiles/r03/pipeline_probes.txt) but a
    #  contract test must not fail on host jitter -- it once did, in the middle of a full suite -- so only sanity is asserted here)
    assert e["pipeline_small"]["runs"] == 300 and 0 < e["pipeline_small"]["ms_per_frame"] < 3.0 * e["pipeline_small"]["ms_per_frame_serial"]
    assert e["sustained"]["launches"] >= 8 and e["sustained"]["dp_symbols_per_s"] > 0.5 * d["value"]
    sm = e["pipeline"]["stage_ms"]
    assert set(sm) == {"generate", "train", "epilogue"} and all(v > 0 for v in sm.values()) and sm["train"] > 0.5 * rf["kernel_ms"]
    c4, c2 = e["configs"]["config4_vaeflex"], e["configs"]["config2_awgn"]
    assert c4["kernel"].startswith("vaeq::dp_wave_kernel<25, 8, ") and c4["value"] > 0 and 0 < c4["flop_frac"] < 1
    assert c2["kernel"].startswith("vaeq::awgn_wave_kernel<25, 8, 3, 1, 350>") and c2["value"] > 0 and 0 < c2["flop_frac"] < 1
    ep = c2["epoch_pipeline"]
    assert ep["ms_train_part"] > 0 and ep["ms_validation_part"] > ep["ms_train_part"] and ep["run_epochs_per_s_epe2"] > 0
    assert ep["validation_forms_agree_bitwise"] and ep["ms_validation_part"] < ep["ms_validation_part_two_step"]   # noise on load: same SER, no noisy frame in HBM
    nn = e["configs"]["vaenn_f3"]                                               # row f3: the VAE-NN training launch and validation pass at the script's shape
    assert nn["unit"] == "symbols/s" and nn["tflops"] > 40.0 and nn["validate_tflops"] > 45.0 and 0.2 < nn["flop_frac"] < 1.0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "2170" in cb["sample"] and cb["reference_dp_symbols_per_s"] == 2170.0


def test_bench_two_ranks_rehearsal():
    """`bench.py --gpus 2` exactly as the driver launches it, with gloo instead of RCCL and both ranks on cuda:0 (VERDICT r1 #7a): weak scaling
    bookkeeping, the gather inside every timed region, the same number of timed regions on both ranks."""
    env = dict(os.environ, VAEQ_DIST_BACKEND="gloo", VAEQ_BENCH_SINGLE_DEVICE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--runs", "256", "--steps", "2", "--warmup", "1",
                        "--min-seconds", "0.2"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["parity"]["ok"]
    assert abs(d["value"] - 2 * d["per_gpu"]) < 1e-9 * d["value"]
    assert abs(d["value"] - 2 * 256 * 10000 * 2 / (d["region_ms"]["median"] * 1e-3)) < 1e-6 * d["value"]
    assert "extra" not in d and "cpu_baseline" not in d                               # N = 1 only
    assert "x2" in d["config"]["parallelism"]


def test_rccl_gather_single_rank():
    """RCCL executes on the one GPU of the box: `bench.py --gpus 1` under torch.distributed.run with VAEQ_FORCE_COLLECTIVE=1 initialises the process group
    with backend nccl (= RCCL) at world size 1 and takes the N > 1 code path -- barrier, all_gather_into_tensor of DEVICE rows inside every timed region
    (sweep.gather_rows), all_reduce(MAX) of the region time -- so the collective branch has run on hardware before an 8-GPU node sees it."""
    env = dict(os.environ, VAEQ_FORCE_COLLECTIVE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("VAEQ_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29641", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--runs", "256", "--steps", "2", "--warmup", "1",
                        "--min-seconds", "0.2", "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 1 and d["parity"]["ok"] and "(nccl)" in d["config"]["parallelism"]
    assert abs(d["value"] - 256 * 10000 * 2 / (d["region_ms"]["median"] * 1e-3)) < 1e-6 * d["value"]


def test_bench_config5_two_ranks_rehearsal():
    """`bench.py --config5 --gpus 2` (strong scaling: a fixed total of 4 nu x 5 SNR x 3 lr x iter runs sharded r mod N, one gather), gloo, both ranks on cuda:0."""
    env = dict(os.environ, VAEQ_DIST_BACKEND="gloo", VAEQ_BENCH_SINGLE_DEVICE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29651", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config5", "--iter", "3", "--steps", "2", "--warmup", "1",
                        "--min-seconds", "0.2"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["parity"]["ok"]
    assert d["config"]["runs_total"] == 180 and d["config"]["runs_per_gpu"] == 90 and "config 5" in d["config"]["workload"]
    assert abs(d["value"] - 180 * 10000 * 2 / (d["region_ms"]["median"] * 1e-3)) < 1e-6 * d["value"]
    assert d["roofline"]["traffic"] is None and "extra" not in d and "cpu_baseline" not in d
