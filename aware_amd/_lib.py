"""ctypes binding of libaware_hip.so (the C ABI declared in include/aware_hip.h).

The shared library is built in-tree by `build_library()` (called from
__graft_entry__.build()).  There is no CPU fallback: every compute entry point of this
package goes through this library and raises if it is missing or if no GPU is present.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AWARE_HIP_LIB") or os.path.join(_HERE, "libaware_hip.so")
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["capi.hip", "dsp_kernels.hip", "dsp_stream.hip", "seam_kernels.hip", "detector_kernels.hip", "gemm_x3.hip", "gemm_h2.hip", "attack_kernels.hip"]

AWARE_OK = 0
ERRORS = {-1: "bad argument", -2: "unsupported configuration", -3: "HIP runtime error", -4: "workspace too small"}


class AwareHipError(RuntimeError):
    pass


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 into aware_amd/libaware_hip.so (hipcc cross-compiles
    without a GPU).  Each source is compiled to an object under csrc/build/ (only those older than
    their inputs, in parallel), then linked."""
    from concurrent.futures import ThreadPoolExecutor
    hdrs = [os.path.join(CSRC, h) for h in ("common.hpp", "fft512.hpp", "kernels.h", "dsp_args.hpp", "split_bf16.hpp")] + [
        os.path.join(_HERE, "..", "include", "aware_hip.h")]
    hdr_t = max(os.path.getmtime(h) for h in hdrs)
    bdir = os.path.join(CSRC, "build")
    os.makedirs(bdir, exist_ok=True)
    flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC"]
    # per-file flags.  The FFT kernels are VALU-issue bound and gfx950 issues a packed-f32 instruction (v_pk_fma_f32 ...)
    # in the time of two scalar ones, so the SLP vectoriser's packing only adds the v_mov traffic that lines registers up
    # in pairs: measured fewer cycles per frame with it off.
    extra = {"dsp_stream.hip": ["-fno-slp-vectorize"]}
    jobs, objs = [], []
    for sname in SOURCES:
        src = os.path.join(CSRC, sname)
        obj = os.path.join(bdir, sname.replace(".hip", ".o"))
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append(["hipcc"] + flags + extra.get(sname, []) + ["-c", src, "-o", obj])
    if not jobs and os.path.exists(LIB_PATH) and all(os.path.getmtime(o) <= os.path.getmtime(LIB_PATH) for o in objs):
        return LIB_PATH

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs)
    return LIB_PATH


class EmbedConfig(C.Structure):
    _fields_ = [("num_iterations", C.c_int), ("tolerance_db", C.c_float), ("loss", C.c_int),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("momentum_decay", C.c_float), ("use_graph", C.c_int), ("conv_pipe", C.c_int), ("readout", C.c_int),
                ("dsp_path", C.c_int), ("l1_weight", C.c_float), ("mel", C.c_int)]


class OptimizerConfig(C.Structure):
    _fields_ = [("kind", C.c_int), ("hyp", C.c_float * 8), ("weight_decay", C.c_double), ("table", C.POINTER(C.c_double)),
                ("plateau", C.c_int), ("patience", C.c_int), ("factor", C.c_double), ("threshold", C.c_double),
                ("min_lr", C.c_double), ("eps", C.c_double), ("lr0", C.c_double)]


_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_pi = C.POINTER(C.c_int)

# name -> (restype, argtypes); mirrors include/aware_hip.h one to one
SIGNATURES = {
    "aware_version": (_i, []),
    "aware_last_hip_error": (C.c_char_p, []),
    "aware_plan_create": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i, _i]),
    "aware_plan_destroy": (None, [_vp]),
    "aware_batch_create": (_i, [C.POINTER(_vp), _i, _pi, _pi]),
    "aware_batch_destroy": (None, [_vp]),
    "aware_batch_total_frames": (_i, [_vp]),
    "aware_batch_total_pooled": (_i, [_vp]),
    "aware_batch_total_out": (_i, [_vp]),
    "aware_batch_out_offset": (_i, [_vp, _i]),
    "aware_batch_out_length": (_i, [_vp, _i]),
    "aware_batch_frames": (_i, [_vp, _i]),
    "aware_batch_scratch_bytes": (_sz, [_vp]),
    "aware_stft": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "aware_istft": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "aware_stft_band": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "aware_stft_bwd": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "aware_istft_bwd": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "aware_detector_backward_workspace_bytes": (_sz, [_vp, _vp]),
    "aware_detector_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "aware_polar_decompose": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "aware_polar_decompose_bwd": (_i, [_vp, _vp, _vp, _vp, _sz, _vp]),
    "aware_polar_assemble": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "aware_polar_assemble_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "aware_waveform_normalize_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "aware_nadam_coefficients": (_i, [_i, _f, _f, _f, _f, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "aware_nadam_clamp_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, C.POINTER(C.c_float), _f, _f, _f, _vp]),
    "aware_detector_train_workspace_bytes": (_sz, [_vp, _vp]),
    "aware_detector_weight_gradients": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_vp), C.POINTER(_vp), _vp, _sz, _vp]),
    "aware_detector_update": (_i, [_vp, _vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "aware_detector_train_gradients": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, C.POINTER(_vp), C.POINTER(_vp), _vp, _sz, _vp]),
    "aware_detector_update_device": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), _vp]),
    "aware_detector_create": (_i, [C.POINTER(_vp), _vp, _vp, _i, _i, _pi, C.POINTER(_vp), C.POINTER(_vp)]),
    "aware_detector_destroy": (None, [_vp]),
    "aware_detect_workspace_bytes": (_sz, [_vp, _vp]),
    "aware_detect": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "aware_detector_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "aware_embed_workspace_bytes": (_sz, [_vp, _vp]),
    "aware_embed_create": (_i, [C.POINTER(_vp), _vp, _vp, _vp, C.POINTER(EmbedConfig), _vp, _sz, _vp]),
    "aware_embed_destroy": (None, [_vp]),
    "aware_embed_set_optimizer": (_i, [_vp, C.POINTER(OptimizerConfig), _vp]),
    "aware_opt_clamp_step": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, C.POINTER(C.c_float), C.POINTER(C.c_float), _vp]),
    "aware_embed_begin": (_i, [_vp, _vp, _vp, _vp]),
    "aware_embed_iterate": (_i, [_vp, _i, _vp]),
    "aware_embed_profile": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "aware_embed_gradient": (_i, [_vp, _vp, _vp]),
    "aware_embed_finish": (_i, [_vp, _vp, _vp, _vp]),
    "aware_embed_buffer": (_vp, [_vp, _i]),
    "aware_pcm_quantize": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "aware_waveform_normalize": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "aware_upfirdn": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _vp]),
    "aware_iir": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "aware_decimate_interp": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "aware_segment_cut": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "aware_phase_vocoder": (_i, [_vp, _vp, _vp, _vp, _i, C.c_double, _vp]),
    "aware_snr": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "aware_gaussian_noise": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _f, _vp, _vp]),
    "aware_spectral_quantize": (_i, [_vp, _i, _f, _f, _vp]),
    "aware_spectral_quantize_bwd": (_i, [_vp, _vp, _vp, _i, _f, _f, _vp]),
    "aware_gemm_nt_variant": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "aware_x3_packed_bytes": (_sz, [_i, _i]),
    "aware_x3_pack": (_i, [_vp, _i, _i, _vp]),
    "aware_gemm_clip": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "aware_gemm_clip_last": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "aware_gemm_clip_h2_workspace_bytes": (_sz, [_i, _i, _i]),
    "aware_gemm_clip_h2": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "aware_gemm_nt": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
}

_lib = None


def load_library():
    """dlopen the in-tree library and attach the prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AwareHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  aware_amd has no CPU fallback.")
    # torch first: PyTorch-ROCm ships its own HIP runtime (libamdhip64); loading it before this library
    # makes both resolve to the same runtime instance, so device pointers and streams are interchangeable
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != AWARE_OK:
        lib = load_library()
        detail = lib.aware_last_hip_error().decode() if rc == -3 else ""
        raise AwareHipError(f"{what}: {ERRORS.get(rc, rc)} {detail}".strip())


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise AwareHipError("aware_amd needs an MI355X (gfx950) GPU: torch.cuda.is_available() is False "
                            "and there is no CPU fallback")
