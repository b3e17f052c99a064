"""Helper for tests/test_processing_gpu.py::test_eval_run_dp_untouched_defaults_are_fast: Eval_run_DP.main() with NO constant touched except
the output directory -- the sweep a user of the reference gets by running the script (15 runs x 170 frames x 10 000 symbols, unseeded)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (import time is the image's, not the sweep's)
from vae_equalizer_amd import Eval_run_DP as ev  # noqa: E402

ev.savePATH = sys.argv[1]
t0 = time.perf_counter()
name, d = ev.main()
print(json.dumps({"seconds": time.perf_counter() - t0, "mat": name, "generator": ev.generator, "base_seed": ev.base_seed}))
