"""Numpy restatement of the STFT / iSTFT conventions on the reference hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows what `torch.stft` / `torch.istft` compute for the reference's calls
  DCCRN/dccrn_decode_vb.py:37-38     torch.stft(x, 512, 128, 512, hann(512))
  FullSubNet/fullsubnet_sa_decode_vb.py:46-47   (512, 256, 512)
  CTSNet/two_stage_com_decode_vb.py:70-71, TaylorSENet/taylorsenet_decode_vb.py:36-37  (320,160,320)
  Uformer/uformer.py:178,182,186,276  (512, 160, win 400)
and, for the librosa front end of LSTM/CRN/GCRN/DPCRN/G2Net
  (e.g. CRN/crn_decode_vb.py:36 `librosa.stft(x, n_fft=320, hop_length=160, window='hanning')`,
   :50-51 `librosa.istft(..., hop_length, win_length, window='hanning', length=L)`),
the same convention: centre=True, reflect padding by n_fft/2, periodic Hann
(`'hanning'` / `torch.hann_window(win)` default), one-sided spectrum, inverse
normalised by the overlap-added squared window.  librosa itself is not
importable in the build container and the reference pins no version, so the
librosa boundary is PARITY UNPINNED; the torch calls are pinned by fixtures.
"""
import numpy as np


def hann_periodic(win_length, dtype=np.float64):
    """torch.hann_window(win_length) (periodic=True) == scipy 'hann' with fftbins=True."""
    n = np.arange(win_length, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)).astype(dtype)


def _padded_window(n_fft, win_length, dtype):
    w = hann_periodic(win_length, dtype)
    if win_length < n_fft:                     # torch centres a short window in n_fft
        left = (n_fft - win_length) // 2
        w = np.pad(w, (left, n_fft - win_length - left))
    return w


def n_frames(length, hop):
    """centre=True frame count for a signal of `length` samples."""
    return 1 + length // hop


def stft(x, n_fft, hop, win_length=None):
    """x [..., L] real -> complex [..., F=n_fft/2+1, T]  (torch.stft layout)."""
    win_length = win_length or n_fft
    x = np.asarray(x)
    dt = x.dtype if x.dtype in (np.float32, np.float64) else np.float64
    w = _padded_window(n_fft, win_length, np.float64)
    pad = n_fft // 2
    xp = np.pad(x.astype(np.float64), [(0, 0)] * (x.ndim - 1) + [(pad, pad)], mode='reflect')
    T = 1 + (xp.shape[-1] - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(T)[:, None]      # [T, n_fft]
    frames = xp[..., idx] * w                                           # [..., T, n_fft]
    spec = np.fft.rfft(frames, n=n_fft, axis=-1)                        # [..., T, F]
    spec = np.swapaxes(spec, -1, -2)
    return spec.astype(np.complex64 if dt == np.float32 else np.complex128)


def istft(spec, n_fft, hop, win_length=None, length=None):
    """complex [..., F, T] -> real [..., L]  (torch.istft, centre=True).
    length=None -> hop*(T-1) samples (torch default, e.g. CTSNet/...vb.py:93,
    Uformer/uformer.py:276); else trimmed / zero-padded to `length`."""
    win_length = win_length or n_fft
    spec = np.asarray(spec)
    w = _padded_window(n_fft, win_length, np.float64)
    T = spec.shape[-1]
    frames = np.fft.irfft(np.swapaxes(spec, -1, -2).astype(np.complex128), n=n_fft, axis=-1) * w   # [..., T, n_fft]
    full = n_fft + hop * (T - 1)
    y = np.zeros(spec.shape[:-2] + (full,), dtype=np.float64)
    env = np.zeros(full, dtype=np.float64)
    for t in range(T):
        y[..., t * hop: t * hop + n_fft] += frames[..., t, :]
        env[t * hop: t * hop + n_fft] += w * w
    start = n_fft // 2
    end = full - n_fft // 2 if length is None else start + length
    y = y[..., start:end]
    env = env[start:end]
    y = np.where(env > 1e-11, y / np.where(env > 1e-11, env, 1.0), y)
    if length is not None and y.shape[-1] < length:
        y = np.pad(y, [(0, 0)] * (y.ndim - 1) + [(0, length - y.shape[-1])])
    out_dt = np.float32 if spec.dtype == np.complex64 else np.float64
    return y.astype(out_dt)


# ----------------------------------------------------------------------------
# decode-script arithmetic shared by every `enhance()`  
# ----------------------------------------------------------------------------
def rms_scale(x):
    """c = sqrt(L / sum x^2)   (e.g. DCCRN/dccrn_decode_vb.py:27, LSTM/lstm_decode_vb.py:35)."""
    x = np.asarray(x, dtype=np.float64)
    return np.sqrt(x.shape[-1] / np.sum(x ** 2.0, axis=-1))


def pad_to_hop(x, n_fft, hop):
    """Tail zero-pad of DCCRN/dccrn_decode_vb.py:32-35 (same code in
    CTSNet/two_stage_com_decode_vb.py:66-69, TaylorSENet/...vb.py:32-35):
    frame_num = ceil((L - win + win)/hop + 1); pad to (frame_num-1)*hop."""
    L = x.shape[-1]
    frame_num = int(np.ceil((L - n_fft + n_fft) / hop + 1))
    fake = (frame_num - 1) * hop + n_fft - n_fft
    return np.concatenate([x, np.zeros(x.shape[:-1] + (fake - L,), dtype=x.dtype)], axis=-1)


def compress_polar(spec, p):
    """|X|**p, angle(X), and the recombined RI pair
    (e.g. DCCRN/dccrn_decode_vb.py:40-42, DPCRN/dpcrn_decode_vb.py:41-45)."""
    mag = np.abs(spec) ** p
    ph = np.angle(spec)
    return mag, ph, mag * np.cos(ph), mag * np.sin(ph)
