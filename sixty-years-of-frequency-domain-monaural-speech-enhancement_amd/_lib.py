"""ctypes binding of libse_engine.so (the C ABI declared in include/se_engine.h).

Fails loudly: there is no CPU / PyTorch fallback for the decode path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SE_ENGINE_LIB") or os.path.join(_HERE, "libse_engine.so")      # SE_ENGINE_LIB: A/B another build

MODEL_IDS = {
    'lstm': 1, 'crn': 2, 'gcrn': 3, 'dpcrn': 4, 'dccrn': 5, 'fullsubnet': 6, 'ctsnet': 7, 'g2net': 8,
    'taylorsenet': 9, 'uformer': 10,
}

# every symbol include/se_engine.h declares
SYMBOLS = [
    'se_abi_version', 'se_last_error', 'se_engine_create', 'se_engine_destroy', 'se_engine_set_tensor',
    'se_engine_finalize', 'se_forward', 'se_enhance_batch', 'se_output_samples', 'se_rms_scale', 'se_stft',
    'se_istft', 'se_num_frames', 'se_num_bins', 'se_set_profiling', 'se_get_profile', 'se_resample',
    'se_resample_samples', 'se_enhance_ragged', 'se_get_stage_profile', 'se_stream_begin', 'se_stream_begin_running', 'se_stream_push', 'se_stream_flush',
    'se_uformer_forward', 'se_pcm16_decode', 'se_pcm16_encode', 'se_frontend', 'se_backend',
]


class SeConfig(C.Structure):
    _fields_ = [('model', C.c_int32), ('device', C.c_int32), ('max_batch', C.c_int32), ('max_samples', C.c_int32),
                ('p_in', C.c_float), ('p_out', C.c_float), ('n_fft', C.c_int32), ('hop', C.c_int32),
                ('win', C.c_int32), ('flags', C.c_int32)]


_lib = None


def load():
    """Load the engine library; raises if it has not been built (`python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP engine is not built (run __graft_entry__.build() or "
            f"`make -C {os.path.join(_HERE, 'csrc')}`). There is no CPU fallback for the decode path.")
    # One HIP runtime per process: the engine shares device memory and streams with torch, so torch's bundled libamdhip64
    # has to be the copy the engine library binds to.  Loading the engine first would pull in /opt/rocm's runtime, torch
    # would then bring its own, and the second one to initialise finds "no ROCm-capable device" (seen with
    # `python __graft_entry__.py smoke`, where build() loads the library before smoke() imports torch).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    lib.se_abi_version.restype = i32
    lib.se_last_error.restype = C.c_char_p
    lib.se_last_error.argtypes = [vp]
    lib.se_engine_create.argtypes = [C.POINTER(SeConfig), C.POINTER(vp)]
    lib.se_engine_destroy.argtypes = [vp]
    lib.se_engine_set_tensor.argtypes = [vp, C.c_char_p, vp, C.POINTER(i64), i32, i32]
    lib.se_engine_finalize.argtypes = [vp]
    lib.se_forward.argtypes = [vp, vp, C.POINTER(i64), i32, vp, vp]
    lib.se_uformer_forward.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, vp, vp]
    lib.se_enhance_batch.argtypes = [vp, vp, i64, i32, i32, vp, i64, vp]
    lib.se_enhance_ragged.argtypes = [vp, vp, i64, i32, C.POINTER(i32), vp, i64, vp]
    lib.se_output_samples.restype = i64
    lib.se_output_samples.argtypes = [vp, i32]
    lib.se_rms_scale.argtypes = [vp, vp, i64, i32, i32, vp, vp]
    lib.se_stft.argtypes = [vp, vp, i64, i32, i32, vp, f32, vp, vp]
    lib.se_istft.argtypes = [vp, vp, i32, i32, vp, vp, i64, i32, vp]
    lib.se_frontend.argtypes = [vp, vp, i64, i32, i32, vp, vp, vp]
    lib.se_backend.argtypes = [vp, i32, vp, vp, i32, i32, vp, vp, i64, i32, vp]
    lib.se_num_frames.restype = i32
    lib.se_num_frames.argtypes = [vp, i32]
    lib.se_num_bins.restype = i32
    lib.se_num_bins.argtypes = [vp]
    lib.se_set_profiling.argtypes = [vp, i32]
    lib.se_get_profile.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double)]
    lib.se_get_stage_profile.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double)]
    lib.se_stream_begin.argtypes = [vp, i32, i32, vp, vp]
    lib.se_stream_begin_running.argtypes = [vp, i32, i32, vp]
    lib.se_stream_push.argtypes = [vp, vp, i64, i32, vp, i64, C.POINTER(i32), vp]
    lib.se_stream_flush.argtypes = [vp, vp, i64, C.POINTER(i32), vp]
    lib.se_resample_samples.restype = i64
    lib.se_resample_samples.argtypes = [i32, i32, i32]
    lib.se_pcm16_decode.argtypes = [vp, i64, i32, i32, vp, i64, vp]
    lib.se_pcm16_encode.argtypes = [vp, i64, i32, i32, vp, i64, vp]
    lib.se_resample.argtypes = [vp, i64, i32, i32, i32, i32, vp, i64, vp]
    _lib = lib
    return lib
