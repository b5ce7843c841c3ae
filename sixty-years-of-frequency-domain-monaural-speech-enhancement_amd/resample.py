"""`librosa.resample(y, orig_sr, target_sr, fix=True, scale=False)` on the GPU (csrc/k_resample.hip through the C ABI).

The reference decode scripts resample every clip to 16 kHz right after reading it (DCCRN/dccrn_decode_vb.py:26,
LSTM/lstm_decode_vb.py:34).  No CPU fallback: the call fails without the HIP library / a GPU.
"""
import ctypes as C

from . import _lib


def resample_samples(n_in, sr_in, sr_out):
    return int(_lib.load().se_resample_samples(int(n_in), int(sr_in), int(sr_out)))


def resample(wav, sr_in, sr_out=16000):
    """wav: float32 cuda tensor [B, L] (or [L]) -> [B, ceil(L * sr_out / sr_in)]."""
    import torch
    lib = _lib.load()
    squeeze = wav.dim() == 1
    x = wav[None] if squeeze else wav
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    if sr_in == sr_out:
        return wav
    B, L = x.shape
    n_out = resample_samples(L, sr_in, sr_out)
    y = torch.empty((B, n_out), dtype=torch.float32, device=x.device)
    st = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    rc = lib.se_resample(C.c_void_p(x.data_ptr()), x.stride(0) if B > 1 else L, B, L, int(sr_in), int(sr_out),
                         C.c_void_p(y.data_ptr()), n_out, st)
    if rc:
        raise RuntimeError(lib.se_last_error(None).decode())
    return y[0] if squeeze else y
