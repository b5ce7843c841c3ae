"""ctypes binding of oracle/_port/libdccrn_cpu.so - the C++ / OpenMP restatement of the DCCRN decode (oracle/dccrn_cpu.cpp).

TEST INFRASTRUCTURE / CPU BASELINE.  Imported only by tests/ and bench.py's `cpu_baseline` leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, '_port', 'libdccrn_cpu.so')
_lib = None


def build():
    subprocess.run(['make', '-C', HERE], check=True, stdout=subprocess.DEVNULL)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        lib = C.CDLL(LIB)
        vp, f32p = C.c_void_p, C.POINTER(C.c_float)
        lib.dccrn_cpu_create.restype = vp
        lib.dccrn_cpu_destroy.argtypes = [vp]
        lib.dccrn_cpu_last_error.restype = C.c_char_p
        lib.dccrn_cpu_set.argtypes = [vp, C.c_char_p, f32p, C.c_long]
        lib.dccrn_cpu_finalize.argtypes = [vp]
        lib.dccrn_cpu_output_samples.restype = C.c_long
        lib.dccrn_cpu_output_samples.argtypes = [C.c_long]
        lib.dccrn_cpu_forward.argtypes = [vp, f32p, C.c_int, f32p, C.c_int]
        lib.dccrn_cpu_enhance.argtypes = [vp, f32p, C.c_long, C.c_int, C.c_long, C.c_float, C.c_float, f32p, C.c_long,
                                          C.c_int, C.c_int]
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class DccrnCpu:
    """The reference's DCCRN (decode script's constructor) + decode loop on the host CPU."""

    def __init__(self, state_dict):
        self.lib = load()
        self.h = C.c_void_p(self.lib.dccrn_cpu_create())
        for k, v in state_dict.items():
            a = np.asarray(v)
            if a.dtype == np.int64:                   # num_batches_tracked
                continue
            a = np.ascontiguousarray(a, dtype=np.float32).ravel()
            self.lib.dccrn_cpu_set(self.h, k.encode(), _p(a), a.size)
        if self.lib.dccrn_cpu_finalize(self.h):
            raise RuntimeError(self.lib.dccrn_cpu_last_error().decode())

    def __del__(self):
        if getattr(self, 'h', None):
            self.lib.dccrn_cpu_destroy(self.h)
            self.h = None

    def forward(self, x, threads=1):
        """x [B,2,257,T] -> [B,2,257,T]  (DCCRN.forward, DCCRN_cprs.py:142-226)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.empty_like(x)
        for b in range(x.shape[0]):
            assert self.lib.dccrn_cpu_forward(self.h, _p(x[b]), x.shape[3], _p(y[b]), threads) == 0
        return y

    def enhance(self, wav, p_in=1.0, p_out=1.0, threads=1, mode=0):
        """wav [n] or [B, n] -> enhanced [padded n] / [B, padded n]  (dccrn_decode_vb.py:25-62).
        mode 0: one clip after the other with `threads` threads inside each layer; mode 1: `threads` clips in flight."""
        w = np.ascontiguousarray(np.atleast_2d(wav), dtype=np.float32)
        B, n = w.shape
        n_out = int(self.lib.dccrn_cpu_output_samples(n))
        out = np.empty((B, n_out), np.float32)
        rc = self.lib.dccrn_cpu_enhance(self.h, _p(w), n, B, n, p_in, p_out, _p(out), n_out, threads, mode)
        if rc:
            raise RuntimeError('dccrn_cpu_enhance failed')
        return out[0] if np.ndim(wav) == 1 else out
