"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's VAE-equalizer hot path (C via ctypes for the
training loop, numpy for the per-frame epilogue).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product package ``vae_equalizer_amd`` never does.

Parity is pinned: ``tests/test_oracle_golden.py`` checks every entry point
against vectors captured from the reference itself (``tools/capture_golden.py``
-> ``tests/golden/*.npz``).
"""
from .capi import *  # noqa: F401,F403
from .epilogue import *  # noqa: F401,F403
