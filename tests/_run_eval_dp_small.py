"""Helper for tests/test_processing_gpu.py::test_eval_run_dp_sharded_two_ranks: a tiny Eval_run_DP sweep under torch.distributed.run."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_equalizer_amd import Eval_run_DP as ev

ev.SNR_vec, ev.lr_optim_vec, ev.iter, ev.num_frames, ev.N_frame_max = [20, 24], [2.5e-3, 2e-3], 2, 2, 400
ev.savePATH, ev.base_seed = sys.argv[1], 5
ev.main()
