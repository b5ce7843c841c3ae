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
