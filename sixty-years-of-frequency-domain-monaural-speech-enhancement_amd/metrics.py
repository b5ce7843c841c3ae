"""Objective scorers for the decode output: STOI, ESTOI and SDR, host side, numpy / scipy.

The reference evaluates with MATLAB files it ships under `DeepXi/deepxi/` (`stoi.m`, `composite.m`, `pesq.m`); nothing
runnable is shipped for Python.  `stoi` below restates `DeepXi/deepxi/stoi.m` line by line (1/3-octave TF units,
384 ms segments, clipping at -15 dB SDR, silent-frame removal with a 40 dB range).  One step is not pinned: stoi.m
resamples to 10 kHz with MATLAB/Octave `resample`, restated here by `scipy.signal.resample_poly` (polyphase FIR, Kaiser
beta 5) - scores agree with the published measure to about the third decimal, and the engine-vs-reference comparison
the tests make (same scorer on both outputs) does not depend on it.  PESQ (`pesq.m`, ITU-T P.862 wide-band) is restated
in `se_amd/pesq.py` (`from se_amd.pesq import pesq`).
"""
import numpy as np

FS, N_FRAME, K_FFT, J_BANDS, MN, N_SEG, BETA, DYN = 10000, 256, 512, 15, 150.0, 30, -15.0, 40.0


def _hanning(n):
    """MATLAB hanning(N): 0.5 (1 - cos(2 pi k / (N + 1))), k = 1..N (no zero end points); stoi.m:146,163."""
    return 0.5 * (1.0 - np.cos(2.0 * np.pi * np.arange(1, n + 1) / (n + 1)))


def _thirdoct(fs, n_fft, num_bands, mn):
    """stoi.m:113-139: rectangular 1/3-octave bands on the FFT grid."""
    f = np.linspace(0, fs, n_fft + 1)[:n_fft // 2 + 1]
    k = np.arange(num_bands)
    fl = np.sqrt((2.0 ** (k / 3.0) * mn) * 2.0 ** ((k - 1) / 3.0) * mn)
    fr = np.sqrt((2.0 ** (k / 3.0) * mn) * 2.0 ** ((k + 1) / 3.0) * mn)
    A = np.zeros((num_bands, len(f)))
    for i in range(num_bands):
        lo = int(np.argmin((f - fl[i]) ** 2))
        hi = int(np.argmin((f - fr[i]) ** 2))
        A[i, lo:hi] = 1.0
    return A


def _remove_silent_frames(x, y, rng, n, k):
    """stoi.m:159-188."""
    starts = np.arange(0, len(x) - n, k)
    w = _hanning(n)
    if len(starts) == 0:
        return x, y
    energy = np.array([20.0 * np.log10(np.linalg.norm(x[s:s + n] * w) / np.sqrt(n) + 1e-300) for s in starts])
    keep = (energy - energy.max() + rng) > 0
    xs, ys = np.zeros_like(x), np.zeros_like(y)
    count, end = 0, 0
    for j, s in enumerate(starts):
        if keep[j]:
            o = starts[count]
            xs[o:o + n] += x[s:s + n] * w
            ys[o:o + n] += y[s:s + n] * w
            end = o + n
            count += 1
    return xs[:end], ys[:end]


def _stdft(x, n, k, n_fft):
    """stoi.m:141-157."""
    starts = np.arange(0, len(x) - n, k)
    w = _hanning(n)
    return np.array([np.fft.fft(x[s:s + n] * w, n_fft) for s in starts])


def stoi(x, y, fs_signal):
    """Short-time objective intelligibility of processed `y` against clean `x` (stoi.m:1-111)."""
    x, y = np.asarray(x, dtype=np.float64).ravel(), np.asarray(y, dtype=np.float64).ravel()
    if len(x) != len(y):
        raise ValueError('x and y should have the same length')
    if fs_signal != FS:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(FS, int(fs_signal))
        x = resample_poly(x, FS // g, int(fs_signal) // g)
        y = resample_poly(y, FS // g, int(fs_signal) // g)
    H = _thirdoct(FS, K_FFT, J_BANDS, MN)
    x, y = _remove_silent_frames(x, y, DYN, N_FRAME, N_FRAME // 2)
    xh = _stdft(x, N_FRAME, N_FRAME // 2, K_FFT)[:, :K_FFT // 2 + 1].T
    yh = _stdft(y, N_FRAME, N_FRAME // 2, K_FFT)[:, :K_FFT // 2 + 1].T
    if xh.shape[1] < N_SEG:
        raise ValueError('not enough non-silent frames for one 384 ms STOI segment')
    X = np.sqrt(H @ np.abs(xh) ** 2)
    Y = np.sqrt(H @ np.abs(yh) ** 2)
    c = 10.0 ** (-BETA / 20.0)
    d = []
    for m in range(N_SEG, X.shape[1] + 1):
        xs, ys = X[:, m - N_SEG:m], Y[:, m - N_SEG:m]
        alpha = np.sqrt(np.sum(xs ** 2, axis=1, keepdims=True) / np.sum(ys ** 2, axis=1, keepdims=True))
        yp = np.minimum(ys * alpha, xs + xs * c)
        xn = xs - xs.mean(axis=1, keepdims=True)
        yn = yp - yp.mean(axis=1, keepdims=True)
        xn /= np.sqrt(np.sum(xn ** 2, axis=1, keepdims=True))
        yn /= np.sqrt(np.sum(yn ** 2, axis=1, keepdims=True))
        d.append(np.sum(xn * yn, axis=1))
    return float(np.mean(d))


def estoi(x, y, fs_signal):
    """Extended STOI (Jensen & Taal, IEEE/ACM TASLP 2016) - the `ESTOI` column of the reference's tables
    (`Figure/t11.jpg`, `t12.jpg`; the reference ships no code for it: `G2Net_VB/Backup.py:17` imports the third-party
    `pystoi`, whose `extended=True` branch this follows).  Same front end as `stoi` (10 kHz, silent-frame removal,
    1/3-octave bands, 384 ms segments); each 15 x 30 segment is mean / norm normalised along time (rows), then along
    frequency (columns), and the score is the mean inner product of the normalised columns - no clipping stage."""
    x, y = np.asarray(x, dtype=np.float64).ravel(), np.asarray(y, dtype=np.float64).ravel()
    if len(x) != len(y):
        raise ValueError('x and y should have the same length')
    if fs_signal != FS:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(FS, int(fs_signal))
        x = resample_poly(x, FS // g, int(fs_signal) // g)
        y = resample_poly(y, FS // g, int(fs_signal) // g)
    H = _thirdoct(FS, K_FFT, J_BANDS, MN)
    x, y = _remove_silent_frames(x, y, DYN, N_FRAME, N_FRAME // 2)
    xh = _stdft(x, N_FRAME, N_FRAME // 2, K_FFT)[:, :K_FFT // 2 + 1].T
    yh = _stdft(y, N_FRAME, N_FRAME // 2, K_FFT)[:, :K_FFT // 2 + 1].T
    if xh.shape[1] < N_SEG:
        raise ValueError('not enough non-silent frames for one 384 ms ESTOI segment')
    X = np.sqrt(H @ np.abs(xh) ** 2)
    Y = np.sqrt(H @ np.abs(yh) ** 2)
    eps = np.finfo(np.float64).eps

    def row_col_normalise(s):
        s = s - s.mean(axis=1, keepdims=True)
        s = s / (np.linalg.norm(s, axis=1, keepdims=True) + eps)
        s = s - s.mean(axis=0, keepdims=True)
        return s / (np.linalg.norm(s, axis=0, keepdims=True) + eps)
    d = []
    for m in range(N_SEG, X.shape[1] + 1):
        xn, yn = row_col_normalise(X[:, m - N_SEG:m]), row_col_normalise(Y[:, m - N_SEG:m])
        d.append(np.sum(xn * yn) / N_SEG)
    return float(np.mean(d))


def sdr(ref, est):
    """Signal-to-distortion ratio in dB, 10 log10(|s|^2 / |s - s_hat|^2) (the SDR column of the reference's tables)."""
    ref, est = np.asarray(ref, dtype=np.float64), np.asarray(est, dtype=np.float64)
    return float(10.0 * np.log10(np.sum(ref ** 2) / np.sum((ref - est) ** 2)))


def si_sdr(ref, est):
    """Scale-invariant SDR in dB."""
    ref, est = np.asarray(ref, dtype=np.float64), np.asarray(est, dtype=np.float64)
    a = np.dot(est, ref) / np.dot(ref, ref)
    return float(10.0 * np.log10(np.sum((a * ref) ** 2) / np.sum((est - a * ref) ** 2)))
