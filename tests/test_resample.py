"""Resampler: `librosa.resample(y, orig_sr, 16000, fix=True, scale=False)` of the decode scripts.
librosa / resampy are absent and unversioned in the reference -> the numpy restatement (oracle/resample.py) is pinned by
signal-level properties on CPU, and the HIP kernel is pinned to the restatement on the GPU."""
import numpy as np
import pytest

import se_amd
from conftest import rms


def test_oracle_resampler_properties():
    from oracle import resample as R
    # output length rule: ceil(n * ratio) (librosa fix=True); the computed part is floor(n * ratio), rest zero
    for n, sr in ((4801, 48000), (4410, 44100), (1000, 8000), (999, 22050)):
        y = R.librosa_resample(np.zeros(n), sr, 16000)
        assert len(y) == int(np.ceil(n * 16000 / sr))
    # a band-limited tone comes out as the same tone at the new rate (resampy's table step truncation gives a small
    # gain error when decimating: < 0.5 %)
    for sr, f0 in ((48000, 440.0), (44100, 1000.0), (8000, 500.0)):
        n = sr // 5
        x = 0.5 * np.sin(2 * np.pi * f0 * np.arange(n) / sr)
        y = R.librosa_resample(x, sr, 16000)
        ref = 0.5 * np.sin(2 * np.pi * f0 * np.arange(len(y)) / 16000)
        assert np.abs(y[300:-300] - ref[300:-300]).max() < 2.5e-3
    # content above the new Nyquist is rejected
    x = 0.5 * np.sin(2 * np.pi * 15000.0 * np.arange(9600) / 48000)
    assert rms(R.librosa_resample(x, 48000, 16000)[300:-300]) < 1e-4
    # identity at equal rates
    x = np.random.default_rng(0).standard_normal(100)
    assert np.array_equal(R.librosa_resample(x, 16000, 16000), x)


@pytest.mark.gpu
@pytest.mark.parametrize('sr_in,n', [(48000, 14403), (44100, 9001), (8000, 3000), (22050, 5000)])
def test_hip_resampler_matches_oracle(sr_in, n):
    import torch
    assert torch.cuda.is_available()
    from se_amd import resample as HR
    from oracle import resample as R
    rng = np.random.default_rng(sr_in)
    x = (0.1 * rng.standard_normal((3, n))).astype(np.float32)
    y = HR.resample(torch.from_numpy(x).cuda(), sr_in, 16000).cpu().numpy()
    assert y.shape == (3, HR.resample_samples(n, sr_in, 16000))
    for b in range(3):
        ref = R.librosa_resample(x[b].astype(np.float64), sr_in, 16000)
        assert y[b].shape == ref.shape
        assert np.abs(y[b] - ref).max() < 2e-7, np.abs(y[b] - ref).max()        # fp32 output of a float64 accumulation


@pytest.mark.gpu
def test_decode_driver_resamples_48k_files(tmp_path):
    """VoiceBank+DEMAND ships at 48 kHz: the driver resamples, decodes at 16 kHz and writes 16 kHz PCM_16."""
    import os
    import types
    from se_amd import decode, wavio, synth, schemas
    from oracle import decode as D, resample as R
    mix, out = str(tmp_path / 'noisy48'), str(tmp_path / 'enh')
    os.makedirs(mix)
    t = np.arange(12000) / 48000.0
    x48 = 0.2 * np.sin(2 * np.pi * 300 * t) * (0.5 + 0.5 * np.sin(2 * np.pi * 3 * t)) + \
        0.01 * np.random.default_rng(3).standard_normal(12000)
    wavio.write_wav_pcm16(os.path.join(mix, 'p232_001.wav'), x48, 48000)
    x48q, fs = wavio.read_wav(os.path.join(mix, 'p232_001.wav'))
    assert fs == 48000
    sd = synth.synth_state_dict(schemas.crn_schema(), 12)
    args = types.SimpleNamespace(mix_file_path=mix, esti_clean_file_path=out, fs=16000)
    assert decode.enhance(args, 'crn', state_dict=sd, max_batch=1) == 1
    y, fs = wavio.read_wav(os.path.join(out, 'p232_001.wav'))
    ref = D.enhance_crn(sd, R.librosa_resample(x48q, 48000, 16000))
    assert fs == 16000 and len(y) == len(ref) == 4000
    assert rms(y - ref) < 1e-4
