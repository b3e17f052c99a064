"""Helper for tests/test_processing_gpu.py::test_eval_run_shaping_vaele_untouched_defaults_are_fast: Eval_run_shaping_vaele.main() with NO
constant touched except the output directory (20 unseeded runs x 500 epochs of 1200 training symbols, 250 validations on 15 000 symbols)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (import time is the image's, not the sweep's)
from vae_equalizer_amd import Eval_run_shaping_vaele as ev  # noqa: E402

ev.savePATH = sys.argv[1]
t0 = time.perf_counter()
name, d = ev.main()
print(json.dumps({"seconds": time.perf_counter() - t0, "mat": name, "generator": ev.generator, "base_seed": ev.base_seed}))
