"""Python handle over the C-ABI engine.  torch is used only for device memory and streams."""
import ctypes as C

import numpy as np

from . import _lib


class EngineError(RuntimeError):
    pass


class Engine:
    """One engine handle = one model replica on one GPU (one per rank)."""

    def __init__(self, model, device=0, max_batch=1, max_samples=64000, p_in=1.0, p_out=1.0,
                 n_fft=0, hop=0, win=0, graphs=False, flags=0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        self.model = model
        cfg = _lib.SeConfig(_lib.MODEL_IDS[model], device, max_batch, max_samples, p_in, p_out, n_fft, hop, win,
                            (1 if graphs else 0) | int(flags))        # SE_CFG_GRAPHS | model-specific SE_CFG_* bits
        if self._lib.se_engine_create(C.byref(cfg), C.byref(self._h)):
            raise EngineError(self._lib.se_last_error(None).decode())
        self.device = device
        self.max_batch, self.max_samples = max_batch, max_samples
        self.finalized = False
        self._stream_batch = 0          # rows of the open frame-online stream (0 = none)

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc:
            raise EngineError(self._lib.se_last_error(self._h).decode())

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            self._lib.se_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        """torch's current stream of THIS engine's device (not of torch's current device)."""
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check_tensor(self, t, what, dim=None):
        import torch
        if not (t.is_cuda and t.dtype == torch.float32):
            raise EngineError(f"{what}: expected a float32 cuda tensor, got {t.dtype} on {t.device}")
        if t.device.index != self.device:
            raise EngineError(f"{what}: tensor on cuda:{t.device.index} but the engine lives on cuda:{self.device}")
        if dim is not None and (t.dim() != dim or t.stride(-1) != 1):
            raise EngineError(f"{what}: expected a {dim}-D tensor with unit inner stride, got shape {tuple(t.shape)} "
                              f"strides {t.stride()}")

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd):
        """Strict load of a flat state dict (numpy arrays or torch tensors), then finalize."""
        for k, v in sd.items():
            a = v.detach().cpu().numpy() if hasattr(v, 'detach') else np.asarray(v)
            if a.dtype == np.int64:
                dt = 1
            else:
                a = np.ascontiguousarray(a, dtype=np.float32)
                dt = 0
            a = np.ascontiguousarray(a)
            shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            self._check(self._lib.se_engine_set_tensor(self._h, k.encode(), a.ctypes.data_as(C.c_void_p), shape,
                                                       a.ndim, dt))
        self._check(self._lib.se_engine_finalize(self._h))
        self.finalized = True

    # ------------------------------------------------------------------ compute (torch tensors on the GPU)
    def forward(self, x, out_shape=None):
        import torch
        self._check_tensor(x, 'forward input')
        assert x.is_contiguous()
        out = torch.empty(out_shape or x.shape, dtype=torch.float32, device=x.device)
        shape = (C.c_int64 * x.dim())(*x.shape)
        self._check(self._lib.se_forward(self._h, C.c_void_p(x.data_ptr()), shape, x.dim(),
                                         C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def uformer_forward(self, inputs, src=None, spectra=True):
        """`model(inputs, src)` of Uformer/uformer.py:172-287: (output, src_out, output_cplx, src_cplx); the source outputs
        are None without `src`, the spectra None with spectra=False."""
        import torch
        self._check_tensor(inputs, 'uformer_forward inputs', 2)
        if not inputs.is_contiguous():
            raise EngineError("uformer_forward inputs: expected dense rows")
        B, L = inputs.shape
        n_out, T, F = self.output_samples(L), self.num_frames(L), self.num_bins()
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=inputs.device)
        out = new(B, n_out)
        out_c = new(B, 2, F, T) if spectra else None
        src_o = src_c = None
        if src is not None:
            self._check_tensor(src, 'uformer_forward src', 2)
            if tuple(src.shape) != (B, L) or not src.is_contiguous():
                raise EngineError(f"uformer_forward src: expected a dense {(B, L)} tensor, got {tuple(src.shape)}")
            src_o = new(B, n_out)
            src_c = new(B, 2, F, T) if spectra else None
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        self._check(self._lib.se_uformer_forward(self._h, ptr(inputs), ptr(src), B, L, ptr(out), ptr(src_o), ptr(out_c),
                                                 ptr(src_c), self._stream()))
        return out, src_o, out_c, src_c

    def output_samples(self, n):
        return int(self._lib.se_output_samples(self._h, n))

    def num_frames(self, n):
        return int(self._lib.se_num_frames(self._h, n))

    def num_bins(self):
        return int(self._lib.se_num_bins(self._h))

    def enhance_batch(self, wav, out=None):
        """wav [B, L] float32 cuda tensor -> [B, output_samples(L)]."""
        import torch
        self._check_tensor(wav, 'enhance_batch input', 2)
        B, L = wav.shape
        n_out = self.output_samples(L)
        if out is None:
            out = torch.empty((B, n_out), dtype=torch.float32, device=wav.device)
        else:
            self._check_tensor(out, 'enhance_batch output', 2)
            if out.shape[0] != B or out.shape[1] < n_out:
                raise EngineError(f"enhance_batch output: need [{B}, >= {n_out}], got {tuple(out.shape)}")
        # the stride of a size-1 dimension is arbitrary in torch / numpy: a single row has pitch L
        in_pitch = wav.stride(0) if B > 1 else L
        out_pitch = out.stride(0) if B > 1 else n_out
        self._check(self._lib.se_enhance_batch(self._h, C.c_void_p(wav.data_ptr()), in_pitch, B, L,
                                               C.c_void_p(out.data_ptr()), out_pitch, self._stream()))
        return out

    def enhance_ragged(self, wav, lengths, out=None):
        """wav [B, >= max(lengths)] float32 cuda tensor, lengths: B sample counts (host ints) ->
        [B, output_samples(max(lengths))]; row b holds output_samples(lengths[b]) samples, then zeros."""
        import torch
        self._check_tensor(wav, 'enhance_ragged input', 2)
        B = wav.shape[0]
        lengths = [int(n) for n in lengths]
        if len(lengths) != B or max(lengths) > wav.shape[1]:
            raise EngineError(f"enhance_ragged: {len(lengths)} lengths (max {max(lengths)}) for a {tuple(wav.shape)} batch")
        n_out = self.output_samples(max(lengths))
        if out is None:
            out = torch.empty((B, n_out), dtype=torch.float32, device=wav.device)
        else:
            self._check_tensor(out, 'enhance_ragged output', 2)
            if out.shape[0] != B or out.shape[1] < n_out:
                raise EngineError(f"enhance_ragged output: need [{B}, >= {n_out}], got {tuple(out.shape)}")
        in_pitch = wav.stride(0) if B > 1 else wav.shape[1]
        out_pitch = out.stride(0) if B > 1 else out.shape[1]
        arr = (C.c_int32 * B)(*lengths)
        self._check(self._lib.se_enhance_ragged(self._h, C.c_void_p(wav.data_ptr()), in_pitch, B, arr,
                                                C.c_void_p(out.data_ptr()), out_pitch, self._stream()))
        return out

    # ------------------------------------------------------------------ frame-online decoding
    def stream_begin(self, batch, c=None, max_chunk_frames=16, running_rms=False):
        """Start `batch` parallel streams; c: per-stream scale tensor (what rms_scale() returns offline) or None = 1.
        running_rms=True: no scale from the caller - the engine keeps a running unit-RMS estimate (se_stream_begin_running)."""
        batch = int(batch)
        if not 1 <= batch <= self.max_batch:
            raise EngineError(f"stream_begin: batch {batch} outside 1..max_batch ({self.max_batch})")
        if c is not None:
            self._check_tensor(c, 'stream_begin c')
            if not c.is_contiguous() or c.numel() < batch:        # the engine copies `batch` floats from c
                raise EngineError(f"stream_begin c: need a contiguous tensor of >= {batch} scales, got shape {tuple(c.shape)}")
        self._stream_batch = 0
        if running_rms:
            if c is not None:
                raise EngineError("stream_begin: running_rms=True and a caller-provided scale exclude each other")
            self._check(self._lib.se_stream_begin_running(self._h, batch, max_chunk_frames, self._stream()))
            self._stream_batch = batch
            return
        self._check(self._lib.se_stream_begin(self._h, batch, max_chunk_frames,
                                              C.c_void_p(c.data_ptr()) if c is not None else None, self._stream()))
        self._stream_batch = batch

    def stream_push(self, wav):
        """wav [batch, n_new] -> the output samples that became final, [batch, n_out] (n_out may be 0)."""
        import torch
        self._check_tensor(wav, 'stream_push input', 2)
        B, n = wav.shape
        if not self._stream_batch:
            raise EngineError("stream_push without stream_begin")
        if B != self._stream_batch:      # the engine reads and writes exactly the stream's rows
            raise EngineError(f"stream_push: {B} rows pushed into a stream of {self._stream_batch}")
        out = torch.empty((B, n + 1024), dtype=torch.float32, device=wav.device)
        n_out = C.c_int32(0)
        pitch = wav.stride(0) if B > 1 else max(n, 1)
        self._check(self._lib.se_stream_push(self._h, C.c_void_p(wav.data_ptr()), pitch, n, C.c_void_p(out.data_ptr()),
                                             out.stride(0), C.byref(n_out), self._stream()))
        return out[:, :n_out.value]

    def stream_flush(self):
        """End of the streams: the remaining output samples, [batch, n_out]."""
        import torch
        if not self._stream_batch:
            raise EngineError("stream_flush without stream_begin")
        out = torch.empty((self._stream_batch, self.max_samples), dtype=torch.float32, device=torch.device('cuda', self.device))
        n_out = C.c_int32(0)
        self._check(self._lib.se_stream_flush(self._h, C.c_void_p(out.data_ptr()), out.stride(0), C.byref(n_out),
                                              self._stream()))
        self._stream_batch = 0
        return out[:, :n_out.value]

    # ------------------------------------------------------------------ stage hooks
    def rms_scale(self, wav):
        import torch
        c = torch.empty(wav.shape[0], dtype=torch.float32, device=wav.device)
        self._check(self._lib.se_rms_scale(self._h, C.c_void_p(wav.data_ptr()), wav.stride(0), wav.shape[0],
                                           wav.shape[1], C.c_void_p(c.data_ptr()), self._stream()))
        return c

    def stft(self, wav, c=None, p_in=1.0):
        import torch
        B, L = wav.shape
        T, F = self.num_frames(L), self.num_bins()
        spec = torch.empty((B, 2, F, T), dtype=torch.float32, device=wav.device)
        self._check(self._lib.se_stft(self._h, C.c_void_p(wav.data_ptr()), wav.stride(0), B, L,
                                      C.c_void_p(c.data_ptr()) if c is not None else None, p_in,
                                      C.c_void_p(spec.data_ptr()), self._stream()))
        return spec

    def istft(self, spec, n_out, c=None):
        import torch
        B, _, F, T = spec.shape
        assert spec.is_contiguous()
        wav = torch.empty((B, n_out), dtype=torch.float32, device=spec.device)
        self._check(self._lib.se_istft(self._h, C.c_void_p(spec.data_ptr()), B, T,
                                       C.c_void_p(c.data_ptr()) if c is not None else None,
                                       C.c_void_p(wav.data_ptr()), wav.stride(0), n_out, self._stream()))
        return wav

    BACKEND_KINDS = {'ri': 0, 'mag': 1, 'cmask': 2}

    def frontend(self, wav):
        """se_frontend: (c [B], spec [B,2,F,T]) - unit-RMS scale, tail pad, STFT and |X|^p_in e^{j angle X} of the decode loop."""
        import torch
        B, L = wav.shape
        T, F = self.num_frames(L), self.num_bins()
        c = torch.empty(B, dtype=torch.float32, device=wav.device)
        spec = torch.empty((B, 2, F, T), dtype=torch.float32, device=wav.device)
        self._check(self._lib.se_frontend(self._h, C.c_void_p(wav.data_ptr()), wav.stride(0), B, L, C.c_void_p(c.data_ptr()),
                                          C.c_void_p(spec.data_ptr()), self._stream()))
        return c, spec

    def backend(self, kind, est, n_out, spec=None, c=None):
        """se_backend: the mask / decompress stage + iSTFT + / c alone.  kind 'ri' (est = estimated spectrum [B,2,F,T]),
        'mag' (est = magnitude [B,F,T], noisy phase from spec), 'cmask' (est = complex ratio mask [B,2,F,T] on spec)."""
        import torch
        assert est.is_contiguous() and (spec is None or spec.is_contiguous())
        B, T = est.shape[0], est.shape[-1]
        wav = torch.empty((B, n_out), dtype=torch.float32, device=est.device)
        self._check(self._lib.se_backend(self._h, self.BACKEND_KINDS[kind], C.c_void_p(est.data_ptr()),
                                         C.c_void_p(spec.data_ptr()) if spec is not None else None, B, T,
                                         C.c_void_p(c.data_ptr()) if c is not None else None, C.c_void_p(wav.data_ptr()),
                                         wav.stride(0), n_out, self._stream()))
        return wav

    # ------------------------------------------------------------------ profiling
    def set_profiling(self, on):
        self._check(self._lib.se_set_profiling(self._h, 1 if on else 0))

    def get_profile(self):
        ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
        self._check(self._lib.se_get_profile(self._h, C.byref(ms), C.byref(n), C.byref(fl)))
        return {'gemm_ms': ms.value, 'gemm_launches': n.value, 'gemm_flops': fl.value}

    STAGES = ('rms', 'stft', 'mask', 'istft')

    def get_stage_profile(self):
        """{stage: {'ms', 'launches', 'bytes'}} of the HBM-bound front / back-end kernels of the last profiled call."""
        out = {}
        for i, name in enumerate(self.STAGES):
            ms, n, by = C.c_double(), C.c_int64(), C.c_double()
            self._check(self._lib.se_get_stage_profile(self._h, i, C.byref(ms), C.byref(n), C.byref(by)))
            out[name] = {'ms': ms.value, 'launches': n.value, 'bytes': by.value}
        return out
