"""ctypes binding of ``libvaeq_hip.so`` (C ABI declared in ``include/vaeq.h``).

There is no CPU fallback: if the library is missing or a call fails, this module raises.
PyTorch is used only to own device memory and streams; every pointer that crosses the
boundary is a raw device pointer.
"""
import ctypes as C
import os
import subprocess

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.environ.get("VAEQ_LIB") or os.path.join(_PKG, "libvaeq_hip.so")   # VAEQ_LIB: A/B builds of the kernels (tools/build_variant.sh)
SOURCES = ["vaeq_dp.hip", "vaeq_dp_wave.hip", "vaeq_dp_wave_mw.hip", "vaeq_dp_wave_mw8.hip", "vaeq_dp_wave_bk.hip", "vaeq_dp_wave_b128.hip", "vaeq_dp_wave_fl.hip", "vaeq_awgn.hip", "vaeq_awgn_wave.hip", "vaeq_misc.hip", "vaeq_nn.hip", "vaeq_cma.hip", "vaeq_epilogue.hip", "vaeq_gen.hip"]
HEADERS = ["vaeq_common.h", "vaeq_wave.h", "vaeq_validate.h", "vaeq_dp_wave_kernel.h", "vaeq_gen_fused.h", "vaeq_epilogue_lds.h", "vaeq_noise.h"]
_LIB = None


class VaeqError(RuntimeError):
    pass


def build(force=False, verbose=False):
    """Compile the HIP sources for gfx950 into ``vae_equalizer_amd/libvaeq_hip.so`` (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(_PKG, "csrc")
    srcs = [os.path.join(csrc, s) for s in SOURCES if os.path.exists(os.path.join(csrc, s))]
    deps = srcs + [os.path.join(csrc, h) for h in HEADERS] + [os.path.join(_ROOT, "include", "vaeq.h")]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(d) <= os.path.getmtime(LIB_PATH) for d in deps):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(_ROOT, "include"), "-I", csrc]
    obj_dir = os.path.join(_PKG, "_obj")
    os.makedirs(obj_dir, exist_ok=True)
    hdr_time = max(os.path.getmtime(d) for d in deps[len(srcs):])

    def compile_one(src):                                      # one translation unit -> one object, rebuilt only when stale
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            cmd = [hipcc, *flags, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        return obj

    from concurrent.futures import ThreadPoolExecutor
    try:
        jobs = len(os.sched_getaffinity(0))
    except AttributeError:
        jobs = os.cpu_count() or 1
    with ThreadPoolExecutor(max(1, min(jobs, len(srcs)))) as pool:  # the translation units compile side by side
        objs = list(pool.map(compile_one, srcs))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-lhipfft", "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


class DPArgs(C.Structure):
    """Mirror of ``struct vaeq_dp_args`` (include/vaeq.h)."""
    _fields_ = [
        ("R", C.c_int32), ("n_frames", C.c_int32), ("steps", C.c_int32), ("B", C.c_int32), ("sps", C.c_int32),
        ("M", C.c_int32), ("n_lev", C.c_int32), ("stride_sym", C.c_int32), ("keep_off", C.c_int32), ("keep_len", C.c_int32),
        ("S", C.c_int64),
        ("rx", C.c_void_p), ("W", C.c_void_p), ("h", C.c_void_p),
        ("adam_mW", C.c_void_p), ("adam_vW", C.c_void_p), ("adam_mh", C.c_void_p), ("adam_vh", C.c_void_p),
        ("step", C.c_void_p), ("amp", C.c_void_p), ("P", C.c_void_p), ("var", C.c_void_p), ("nu_sc", C.c_void_p),
        ("lr_W", C.c_void_p), ("lr_h", C.c_void_p),
        ("q_out", C.c_void_p), ("y_out", C.c_void_p), ("loss", C.c_void_p), ("var_est", C.c_void_p),
        ("eq_out", C.c_void_p), ("dec_out", C.c_void_p),
        ("dbg_gW", C.c_void_p), ("dbg_gh", C.c_void_p),
        ("threads", C.c_int32), ("no_update", C.c_int32),
    ]


class AWGNArgs(C.Structure):
    """Mirror of ``struct vaeq_awgn_args`` (include/vaeq.h)."""
    _fields_ = [
        ("R", C.c_int32), ("steps", C.c_int32), ("B", C.c_int32), ("sps", C.c_int32), ("M", C.c_int32), ("n_lev", C.c_int32),
        ("S", C.c_int64),
        ("rx", C.c_void_p), ("W", C.c_void_p), ("h", C.c_void_p),
        ("adam_mW", C.c_void_p), ("adam_vW", C.c_void_p), ("adam_xW", C.c_void_p),
        ("adam_mh", C.c_void_p), ("adam_vh", C.c_void_p), ("adam_xh", C.c_void_p),
        ("step", C.c_void_p), ("amp", C.c_void_p), ("P", C.c_void_p), ("amp_mean", C.c_void_p), ("var", C.c_void_p),
        ("lr", C.c_void_p),
        ("q_out", C.c_void_p), ("y_out", C.c_void_p), ("loss", C.c_void_p),
        ("dbg_gW", C.c_void_p), ("dbg_gh", C.c_void_p),
        ("threads", C.c_int32), ("no_update", C.c_int32),
    ]


class NNArgs(C.Structure):
    """Mirror of ``struct vaeq_nn_args`` (include/vaeq.h)."""
    _fields_ = [
        ("R", C.c_int32), ("steps", C.c_int32), ("B", C.c_int32), ("sps", C.c_int32), ("M", C.c_int32), ("n_lev", C.c_int32),
        ("k1", C.c_int32), ("k2", C.c_int32), ("S", C.c_int64),
        ("rx", C.c_void_p), ("theta", C.c_void_p), ("adam_m", C.c_void_p), ("adam_v", C.c_void_p), ("adam_x", C.c_void_p),
        ("step", C.c_void_p), ("amp", C.c_void_p), ("lr", C.c_void_p), ("loss", C.c_void_p), ("q_out", C.c_void_p), ("dbg_g", C.c_void_p),
        ("no_update", C.c_int32), ("batch_norm", C.c_int32), ("bn_running", C.c_void_p),
    ]


# every symbol include/vaeq.h declares; tests check the library exports all of them
EXPORTS = ["vaeq_dp_train", "vaeq_dp_step_debug", "vaeq_dp_lds_bytes", "vaeq_dp_resident_runs", "vaeq_soft_demap", "vaeq_dp_forward", "vaeq_dp_loss", "vaeq_dp_loss_bwd", "vaeq_dp_forward_bwd", "vaeq_dp_epilogue", "vaeq_dp_epilogue_ws_bytes", "vaeq_dp_epilogue_compact", "vaeq_gen_dp_tx", "vaeq_gen_dp_disperse", "vaeq_gen_dp_finish", "vaeq_gen_dp_frame", "vaeq_awgn_train",
           "vaeq_awgn_lds_bytes", "vaeq_awgn_forward", "vaeq_awgn_validate", "vaeq_awgn_validate_gen", "vaeq_gen_awgn_clean", "vaeq_awgn_loss", "vaeq_awgn_loss_bwd", "vaeq_awgn_forward_bwd", "vaeq_gen_awgn", "vaeq_nn_train", "vaeq_nn_param_count", "vaeq_nn_lds_bytes", "vaeq_nn_forward", "vaeq_nn_validate", "vaeq_cma", "vaeq_cpe", "vaeq_version", "vaeq_strerror", "vaeq_last_kernel", "vaeq_stream_copy", "vaeq_gen_dp_power_parts", "vaeq_cma_epilogue"]


def lib():
    """Load the library (once).  Raises VaeqError if it has not been built -- no silent fallback."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise VaeqError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        L.vaeq_strerror.restype = C.c_char_p
        L.vaeq_strerror.argtypes = [C.c_int]
        L.vaeq_version.restype = C.c_int
        L.vaeq_last_kernel.restype = C.c_int
        L.vaeq_last_kernel.argtypes = [C.c_char_p, C.c_int32]
        L.vaeq_stream_copy.restype = C.c_int
        L.vaeq_stream_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.vaeq_dp_train.restype = C.c_int
        L.vaeq_dp_train.argtypes = [C.POINTER(DPArgs), C.c_void_p]
        L.vaeq_dp_step_debug.restype = C.c_int
        L.vaeq_dp_step_debug.argtypes = [C.POINTER(DPArgs), C.c_void_p, C.c_void_p, C.c_void_p]
        L.vaeq_dp_lds_bytes.restype = C.c_int64
        L.vaeq_dp_lds_bytes.argtypes = [C.c_int32] * 4
        L.vaeq_dp_resident_runs.restype = C.c_int64
        L.vaeq_dp_resident_runs.argtypes = [C.c_int32] * 5
        L.vaeq_soft_demap.restype = C.c_int
        L.vaeq_soft_demap.argtypes = [C.c_int32, C.c_int64, C.c_int32] + [C.c_void_p] * 6
        L.vaeq_dp_forward.restype = C.c_int
        L.vaeq_dp_forward.argtypes = [C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 8
        L.vaeq_dp_loss.restype = C.c_int
        L.vaeq_dp_loss.argtypes = [C.c_int32] * 5 + [C.c_void_p] * 8
        L.vaeq_dp_loss_bwd.restype = C.c_int
        L.vaeq_dp_loss_bwd.argtypes = [C.c_int32] * 5 + [C.c_void_p] * 9
        L.vaeq_dp_forward_bwd.restype = C.c_int
        L.vaeq_dp_forward_bwd.argtypes = [C.c_int32] * 5 + [C.c_void_p] * 9
        L.vaeq_dp_epilogue.restype = C.c_int
        L.vaeq_dp_epilogue.argtypes = [C.c_int32, C.c_int64, C.c_int32, C.c_int32] + [C.c_void_p] * 11
        L.vaeq_cma_epilogue.restype = C.c_int
        L.vaeq_cma_epilogue.argtypes = [C.c_int32, C.c_int64, C.c_int32] + [C.c_void_p] * 10
        L.vaeq_dp_epilogue_compact.restype = C.c_int
        L.vaeq_dp_epilogue_compact.argtypes = [C.c_int32, C.c_int64, C.c_int32, C.c_int32] + [C.c_void_p] * 11
        L.vaeq_dp_epilogue_ws_bytes.restype = C.c_int64
        L.vaeq_dp_epilogue_ws_bytes.argtypes = [C.c_int32, C.c_int64]
        L.vaeq_gen_dp_tx.restype = C.c_int
        L.vaeq_gen_dp_tx.argtypes = [C.c_int32] * 9 + [C.c_void_p] * 3 + [C.c_uint64, C.c_uint32] + [C.c_void_p] * 3
        L.vaeq_gen_dp_disperse.restype = C.c_int
        L.vaeq_gen_dp_disperse.argtypes = [C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double] + [C.c_float] * 5 + [C.c_void_p] * 3
        L.vaeq_gen_dp_finish.restype = C.c_int
        L.vaeq_gen_dp_finish.argtypes = [C.c_int32] * 5 + [C.c_void_p, C.c_uint64, C.c_uint32] + [C.c_void_p] * 5
        if hasattr(L, "vaeq_awgn_train"):
            L.vaeq_awgn_train.restype = C.c_int
            L.vaeq_awgn_train.argtypes = [C.POINTER(AWGNArgs), C.c_void_p]
            L.vaeq_awgn_lds_bytes.restype = C.c_int64
            L.vaeq_awgn_lds_bytes.argtypes = [C.c_int32] * 4
            L.vaeq_awgn_forward.restype = C.c_int
            L.vaeq_awgn_forward.argtypes = [C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 8
        L.vaeq_nn_train.restype = C.c_int
        L.vaeq_nn_train.argtypes = [C.POINTER(NNArgs), C.c_void_p]
        L.vaeq_nn_param_count.restype = C.c_int64
        L.vaeq_nn_param_count.argtypes = [C.c_int32] * 5
        L.vaeq_nn_lds_bytes.restype = C.c_int64
        L.vaeq_nn_lds_bytes.argtypes = [C.c_int32] * 7
        L.vaeq_nn_forward.restype = C.c_int
        L.vaeq_nn_forward.argtypes = [C.c_int32, C.c_int64] + [C.c_int32] * 5 + [C.c_void_p] * 5
        L.vaeq_cma.restype = C.c_int
        L.vaeq_cma.argtypes = [C.c_int32, C.c_int64] + [C.c_int32] * 5 + [C.c_void_p, C.c_float] + [C.c_void_p] * 5
        L.vaeq_cpe.restype = C.c_int
        L.vaeq_cpe.argtypes = [C.c_int32, C.c_int64, C.c_int32] + [C.c_void_p] * 3
        L.vaeq_awgn_loss.restype = C.c_int
        L.vaeq_awgn_loss.argtypes = [C.c_int32] * 5 + [C.c_void_p] * 7
        L.vaeq_awgn_loss_bwd.restype = C.c_int
        L.vaeq_awgn_loss_bwd.argtypes = [C.c_int32] * 5 + [C.c_void_p] * 9
        L.vaeq_awgn_forward_bwd.restype = C.c_int
        L.vaeq_awgn_forward_bwd.argtypes = [C.c_int32] * 5 + [C.c_void_p] * 9
        L.vaeq_nn_validate.restype = C.c_int
        L.vaeq_nn_validate.argtypes = [C.c_int32, C.c_int64] + [C.c_int32] * 6 + [C.c_void_p] * 8
        L.vaeq_gen_dp_power_parts.restype = C.c_int32
        L.vaeq_gen_dp_power_parts.argtypes = [C.c_int32]
        L.vaeq_gen_dp_frame.restype = C.c_int
        L.vaeq_gen_dp_frame.argtypes = ([C.c_int32] * 9 + [C.c_void_p] * 5 + [C.c_double] * 3 + [C.c_float] * 4 + [C.c_uint64, C.c_uint32]
                                        + [C.c_void_p] * 6)
        L.vaeq_awgn_validate.restype = C.c_int
        L.vaeq_awgn_validate.argtypes = [C.c_int32, C.c_int64] + [C.c_int32] * 4 + [C.c_void_p] * 10
        L.vaeq_gen_awgn_clean.restype = C.c_int
        L.vaeq_gen_awgn_clean.argtypes = [C.c_int32] * 8 + [C.c_void_p] * 3 + [C.c_uint64, C.c_uint32] + [C.c_void_p] * 4
        L.vaeq_awgn_validate_gen.restype = C.c_int
        L.vaeq_awgn_validate_gen.argtypes = ([C.c_int32, C.c_int64] + [C.c_int32] * 4 + [C.c_void_p, C.c_int32] + [C.c_void_p] * 3 +
                                             [C.c_uint64, C.c_uint32] + [C.c_void_p] * 10)
        L.vaeq_gen_awgn.restype = C.c_int
        L.vaeq_gen_awgn.argtypes = [C.c_int32] * 8 + [C.c_void_p] * 4 + [C.c_uint64, C.c_uint32] + [C.c_void_p] * 7
        _LIB = L
    return _LIB


def last_kernel():
    """Name of the kernel instantiation this thread's latest vaeq_dp_train / vaeq_awgn_train call launched."""
    buf = C.create_string_buffer(160)
    check(lib().vaeq_last_kernel(buf, 160), "vaeq_last_kernel")
    return buf.value.decode()


def check(code, what):
    if code != 0:
        raise VaeqError(f"{what} failed: {lib().vaeq_strerror(int(code)).decode()} (code {int(code)})")


def ptr(t, dtype=torch.float32):
    """Raw device pointer of a contiguous CUDA(HIP) tensor of the expected dtype; None passes NULL."""
    if t is None:
        return None
    if not t.is_cuda:
        raise VaeqError("the vaeq kernels take device tensors (no CPU path)")
    if t.dtype != dtype or not t.is_contiguous():
        raise VaeqError(f"expected a contiguous {dtype} tensor, got {t.dtype} contiguous={t.is_contiguous()}")
    return C.c_void_p(t.data_ptr())


def current_stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
