"""vae_equalizer_amd -- MI355X-native (gfx950) hot path of kit-cel/vae-equalizer.

Only what the path needs: ``csrc/`` (HIP kernels + C ABI -> ``libvaeq_hip.so``), the ctypes binding
(``_native``), the batched run engines (``engine``) and host-side mirrors of the reference's call surface
(``shared_funcs``, ``func_VAELE_DP_MQAM_shaping``, ``func_VAEflex_DP_MQAM_shaping``, ``func_VAELE_MQAM_shaping``,
``Eval_run_DP``, ``Eval_run_shaping_vaele``).  There is no CPU fallback: without the HIP library every
compute entry point raises.
"""
from . import _native  # noqa: F401

__all__ = ["_native"]
