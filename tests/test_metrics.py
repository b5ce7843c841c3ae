"""Scorers: STOI restated from DeepXi/deepxi/stoi.m, SDR.  CPU: measure-level properties;
GPU: the north-star claim "STOI identical to 3 d.p." - the same scorer on the engine's and the oracle's output."""
import numpy as np
import pytest

import se_amd
from se_amd import metrics, synth


def _speechlike(seed, n=32000):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    x = np.zeros(n)
    for f0 in (140.0, 190.0):
        env = (0.5 + 0.5 * np.sin(2 * np.pi * 3.1 * t + rng.uniform(0, 6))) ** 2
        x += env * sum(np.sin(2 * np.pi * f0 * k * t + rng.uniform(0, 6)) / k for k in range(1, 20))
    return 0.1 * x / np.abs(x).max()


def test_stoi_properties():
    x = _speechlike(0)
    rng = np.random.default_rng(1)
    assert abs(metrics.stoi(x, x, 16000) - 1.0) < 1e-9
    prev = 1.0
    for snr in (20, 10, 0, -10):
        noise = rng.standard_normal(len(x))
        noise *= np.sqrt(np.sum(x ** 2) / np.sum(noise ** 2)) * 10 ** (-snr / 20)
        d = metrics.stoi(x, x + noise, 16000)
        assert 0.0 < d < prev, (snr, d, prev)         # monotone in SNR
        prev = d
    assert metrics.stoi(x, 3.7 * x, 16000) > 0.999    # level invariant
    assert abs(metrics.stoi(x, x, 10000) - 1.0) < 1e-9


def test_estoi_properties():
    """ESTOI (the column the reference's tables report): 1 for identical signals, monotone in SNR, level invariant, and
    below STOI for noisy speech at low SNR (no clipping stage, column normalisation)."""
    x = _speechlike(0)
    rng = np.random.default_rng(1)
    assert abs(metrics.estoi(x, x, 16000) - 1.0) < 1e-9
    prev = 1.0
    for snr in (20, 10, 0, -10):
        noise = rng.standard_normal(len(x))
        noise *= np.sqrt(np.sum(x ** 2) / np.sum(noise ** 2)) * 10 ** (-snr / 20)
        d = metrics.estoi(x, x + noise, 16000)
        assert -0.1 < d < prev, (snr, d, prev)
        prev = d
    assert metrics.estoi(x, 3.7 * x, 16000) > 0.999
    noise = rng.standard_normal(len(x))
    noise *= np.sqrt(np.sum(x ** 2) / np.sum(noise ** 2))
    assert metrics.estoi(x, x + noise, 16000) < metrics.stoi(x, x + noise, 16000)


def _utterances(seed, n=64000, fs=16000):
    """Speech-like test signal for PESQ: drifting pitch, formant-weighted harmonics, syllables of 250-600 ms with pauses
    (the perfectly periodic `_speechlike` is adversarial for P.862's VAD / delay search)."""
    rng = np.random.default_rng(seed)
    f0 = np.clip(160 + 40 * np.cumsum(rng.standard_normal(n)) / np.sqrt(n) * 3, 90, 260)
    ph = 2 * np.pi * np.cumsum(f0) / fs
    x = np.zeros(n)
    for k in range(1, 30):
        x += (1.0 / k) * np.exp(-0.5 * ((k * f0 - rng.uniform(400, 2500)) / 900.0) ** 2) * np.sin(k * ph + rng.uniform(0, 6))
    env, pos = np.zeros(n), int(0.2 * fs)
    while pos < n - int(0.3 * fs):
        on, off = int(rng.uniform(0.25, 0.6) * fs), int(rng.uniform(0.05, 0.2) * fs)
        w = np.hanning(on) ** 0.5 * rng.uniform(0.5, 1.0)
        env[pos:pos + on] = np.maximum(env[pos:pos + on], w[:min(on, n - pos)])
        pos += on + off
    x = x * env + 0.002 * rng.standard_normal(n)
    return 0.2 * x / np.abs(x).max()


def test_pesq_anchors():
    """se_amd/pesq.py (restated from DeepXi/deepxi/pesq.m, unpinned - no vector in the reference): the anchors of P.862."""
    from se_amd import pesq as P
    x = _utterances(3)
    assert abs(P.pesq_raw(x, x) - 4.5) < 1e-9 and abs(P.pesq(x, x) - 4.644) < 1e-3      # identical: raw 4.5, MOS-LQO 4.64
    assert abs(P.pesq_raw(x, 0.3 * x) - 4.5) < 1e-6                                     # level aligned away
    rng = np.random.default_rng(1)
    prev = 4.5
    for snr in (40, 30, 20, 10, 0):
        noise = rng.standard_normal(len(x))
        noise *= np.sqrt(np.sum(x ** 2) / np.sum(noise ** 2)) * 10 ** (-snr / 20)
        r = P.pesq_raw(x, x + noise)
        assert r < prev, (snr, r, prev)                                                 # monotone in SNR
        prev = r
    assert prev < 1.0 and 1.0 < P.pesq(x, x + noise) < 1.3
    delayed = np.concatenate([np.zeros(320), x[:-320]])                                 # 20 ms constant delay is found
    assert P.pesq_raw(x, delayed) > 4.3
    with pytest.raises(ValueError):
        P.pesq(x[::2], x[::2], 8000)


def test_sdr_values():
    x = _speechlike(2)
    n = np.random.default_rng(3).standard_normal(len(x))
    n *= np.sqrt(np.sum(x ** 2) / np.sum(n ** 2)) * 10 ** (-10 / 20)
    assert abs(metrics.sdr(x, x + n) - 10.0) < 1e-9
    assert abs(metrics.si_sdr(x, 2.0 * (x + n)) - metrics.si_sdr(x, x + n)) < 1e-9


@pytest.mark.gpu
def test_engine_and_oracle_outputs_score_identically():
    import torch
    assert torch.cuda.is_available()
    from se_amd.models import crn_net
    from oracle import decode as D
    clean = _speechlike(5)
    noisy = (clean + 0.03 * np.random.default_rng(6).standard_normal(len(clean))).astype(np.float32)
    m = crn_net(max_batch=1, max_samples=len(noisy)).load_synthetic(12)
    sd = synth.synth_state_dict(m.state_dict_schema(), 12)
    y = m.enhance_batch(torch.from_numpy(noisy[None]).cuda()).cpu().numpy()[0].astype(np.float64)
    ref = D.enhance_crn(sd, noisy.astype(np.float64))
    s_eng, s_ref = metrics.stoi(clean, y, 16000), metrics.stoi(clean, ref, 16000)
    assert round(s_eng, 3) == round(s_ref, 3) and abs(s_eng - s_ref) < 1e-6, (s_eng, s_ref)
    assert abs(metrics.sdr(clean, y) - metrics.sdr(clean, ref)) < 1e-4
    e_eng, e_ref = metrics.estoi(clean, y, 16000), metrics.estoi(clean, ref, 16000)
    assert round(e_eng, 3) == round(e_ref, 3) and abs(e_eng - e_ref) < 1e-6, (e_eng, e_ref)


@pytest.mark.gpu
def test_pesq_of_engine_and_reference_path_outputs():
    """BASELINE's quality gate: PESQ of the engine's output within +-0.01 of the reference path's (here: the oracle's) on
    the same noisy input - evaluated, not argued: DPCRN with the reference's REAL checkpoint enhancing a noisy speech-like
    clip, both outputs scored against the clean signal with the restated P.862 wide-band measure."""
    import torch
    from se_amd import pesq as P
    from se_amd.models import dpcrn
    from oracle import decode as D
    from conftest import load_golden
    clean = _utterances(7)
    noisy = (clean + 0.02 * np.random.default_rng(8).standard_normal(len(clean))).astype(np.float32)
    ck = dict(load_golden('ckpt_vb_dpcrn_noncprs'))
    m = dpcrn(max_batch=1, max_samples=len(noisy))
    m.load_state_dict(ck)
    y = m.enhance_batch(torch.from_numpy(noisy[None]).cuda()).cpu().numpy()[0].astype(np.float64)
    ref = D.enhance_dpcrn(ck, noisy.astype(np.float64))
    p_noisy, p_eng, p_ref = P.pesq(clean, noisy), P.pesq(clean, y), P.pesq(clean, ref)
    print('wb-PESQ noisy %.3f  engine %.3f  reference path %.3f' % (p_noisy, p_eng, p_ref))
    assert abs(p_eng - p_ref) <= 0.01 and round(p_eng, 2) == round(p_ref, 2)
    # the file-level path: PCM_16 quantisation of both does not move the score either
    q = lambda v: np.clip(np.rint(v * 32767.0), -32768, 32767) / 32768.0       # sf.write -> sf.read
    assert abs(P.pesq(clean, q(y)) - P.pesq(clean, q(ref))) <= 0.01
