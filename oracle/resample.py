"""Band-limited sinc resampler: numpy restatement of `librosa.resample(y, orig_sr, target_sr, fix=True, scale=False)`
as the decode scripts call it (DCCRN/dccrn_decode_vb.py:26, LSTM/lstm_decode_vb.py:34: VoiceBank+DEMAND ships at 48 kHz).

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: the arithmetic lives in third-party packages that are
absent here and unversioned in the reference (old librosa -> resampy, filter 'kaiser_best').  This file restates the
published resampy algorithm: a Kaiser-windowed sinc table (64 zero crossings, 2^9 samples per crossing, roll-off
0.9475937167399596, beta 14.769656459379492) read with linear interpolation between table entries, accumulated in
float64, output length floor(n * ratio), then librosa's `fix_length` to ceil(n * ratio).
"""
import numpy as np

NUM_ZEROS, PRECISION = 64, 9
ROLLOFF, BETA = 0.9475937167399596, 14.769656459379492


def sinc_window(num_zeros=NUM_ZEROS, precision=PRECISION, rolloff=ROLLOFF, beta=BETA):
    """Right half of the interpolation filter: rolloff * sinc(rolloff * t) * kaiser, t in [0, num_zeros]."""
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    taper = np.kaiser(2 * n + 1, beta)[n:]
    return taper * sinc_win, num_bits


def time_registers(n_out, sample_ratio):
    """The running read position of resampy's loop: sequential float64 accumulation of 1 / ratio."""
    inc = 1.0 / sample_ratio
    t = np.empty(n_out, dtype=np.float64)
    acc = 0.0
    for i in range(n_out):
        t[i] = acc
        acc += inc
    return t


def resample(x, sr_orig, sr_new):
    """resampy.resample(x, sr_orig, sr_new, filter='kaiser_best') for a 1-D float64 signal."""
    x = np.asarray(x, dtype=np.float64)
    ratio = float(sr_new) / sr_orig
    n_out = int(x.shape[0] * ratio)
    win, num_table = sinc_window()
    if ratio < 1:
        win = win * ratio
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    scale = min(1.0, ratio)
    index_step = int(scale * num_table)
    nwin, n_orig = win.shape[0], x.shape[0]
    treg = time_registers(n_out, ratio)
    y = np.zeros(n_out, dtype=np.float64)
    for t in range(n_out):
        n = int(treg[t])
        frac = scale * (treg[t] - n)
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        i_max = min(n + 1, (nwin - offset) // index_step)
        idx = offset + index_step * np.arange(i_max)
        y[t] += np.dot(win[idx] + eta * delta[idx], x[n - np.arange(i_max)])
        frac = scale - frac
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        k_max = min(n_orig - n - 1, (nwin - offset) // index_step)
        idx = offset + index_step * np.arange(k_max)
        y[t] += np.dot(win[idx] + eta * delta[idx], x[n + 1 + np.arange(k_max)])
    return y


def librosa_resample(y, orig_sr, target_sr):
    """librosa.resample(y, orig_sr, target_sr, fix=True, scale=False): length forced to ceil(n * ratio)."""
    y = np.asarray(y, dtype=np.float64)
    if orig_sr == target_sr:
        return y
    n_samples = int(np.ceil(y.shape[-1] * float(target_sr) / orig_sr))
    out = resample(y, orig_sr, target_sr)
    if out.shape[0] < n_samples:
        out = np.concatenate([out, np.zeros(n_samples - out.shape[0])])
    return out[:n_samples]
