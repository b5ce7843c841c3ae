"""GPU: se_enhance_ragged - one call decodes clips of DIFFERENT lengths, each row exactly as if decoded alone (the
reference decodes one clip at a time, `for file_id in file_list`, e.g. DCCRN/dccrn_decode_vb.py:24; VoiceBank+DEMAND has
~824 distinct lengths).  Zero-padding to a common length is NOT neutral for the InstanceNorm / utterance-mean models
(CTSNet step 1 moves by O(1) on the original frames), so rows carry their own length through the unit-RMS
scale, the STFT's reflect padding, every utterance-wide statistic and the iSTFT.

Each row is compared with (a) the same engine decoding that clip alone at its exact length - itself pinned to the
reference fixtures / the oracle by the other GPU tests - and (b), for a few rows, the numpy oracle's per-clip decode.
Bar: 1e-4 RMS on the waveform (north star)."""
import numpy as np
import pytest

import se_amd  # noqa: F401
from se_amd import synth
from conftest import rms

pytestmark = pytest.mark.gpu
WSEED = {'uformer': 21, 'lstm': 11, 'crn': 12, 'dpcrn': 13, 'dccrn': 14, 'fullsubnet': 15, 'gcrn': 16, 'taylorsenet': 19, 'g2net': 20,
         'taylorsenet_new': 19, 'g2net_new': 20}


def _make(name, max_batch, max_samples, **kw):
    from se_amd import models, models_new  # noqa: F401
    if name.startswith('ctsnet'):
        cls = models_new.CTSNet if name.endswith('_new') else models.CTSNet
        return cls(max_batch=max_batch, max_samples=max_samples, **kw).load_synthetic(17, 18)
    return models.MODEL_CLASSES[name](max_batch=max_batch, max_samples=max_samples, **kw).load_synthetic(WSEED[name])


def _oracle(name, m, x):
    from oracle import decode as D
    if name.startswith('ctsnet'):
        tag = '_new' if name.endswith('_new') else ''
        from conftest import load_schema
        sd1 = synth.synth_state_dict(load_schema('cts_step1' + tag), 17)
        sd2 = synth.synth_state_dict(load_schema('cts_step2' + tag), 18)
        return D.enhance_ctsnet(sd1, sd2, x, 0.5, 2.0)
    sd = synth.synth_state_dict(m.state_dict_schema(), WSEED[name])
    if name == 'uformer':                       # STFT / iSTFT inside the model, no exponents in its decode script
        return D.enhance_uformer(sd, x)
    return D.ENHANCE[name.replace('_new', '')](sd, x, 0.5, 2.0)


def _ragged_batch(lengths, seed0):
    kinds = ('speech', 'white', 'speech', 'gap')
    clips = [synth.synth_clip(seed0 + i, kinds[i % 4], n) for i, n in enumerate(lengths)]
    x = np.zeros((len(lengths), max(lengths)), np.float32)
    for i, c in enumerate(clips):
        x[i, :len(c)] = c
        x[i, len(c):] = 0.25 * np.sin(0.01 * np.arange(max(lengths) - len(c)))     # junk past the end must be ignored
    return clips, x


@pytest.mark.parametrize('name', ['dccrn', 'crn', 'lstm', 'gcrn', 'dpcrn', 'fullsubnet', 'ctsnet', 'g2net', 'taylorsenet',
                                  'ctsnet_new', 'taylorsenet_new', 'g2net_new', 'uformer'])
def test_ragged_rows_equal_per_clip_decodes(name):
    import torch
    lengths = [6000, 3217, 9000, 4801, 7777, 5120, 8191, 3999]
    clips, x = _ragged_batch(lengths, 1300)
    kw = {} if name == 'uformer' else dict(p_in=0.5, p_out=2.0)
    m = _make(name, len(lengths), max(lengths), **kw)
    y = m.enhance_ragged(torch.from_numpy(x).cuda(), lengths).cpu().numpy()
    one = _make(name, 1, max(lengths), **kw)
    for i, c in enumerate(clips):
        ref = one.enhance_batch(torch.from_numpy(c[None]).cuda()).cpu().numpy()[0]
        n = len(ref)
        assert n == m.engine.output_samples(lengths[i])
        e = rms(y[i, :n] - ref)
        assert np.isfinite(y[i]).all()
        assert e < 1e-4 and e < 2e-5 * max(rms(ref), 1e-3), (name, i, lengths[i], e, rms(ref))
        assert not y[i, n:].any(), (name, i, 'samples past the row\'s own output length must be zero')
    for i in (1, 4):                                        # the numpy oracle's decode of the clip alone
        ref = _oracle(name, m, clips[i])
        e = rms(y[i, :len(ref)] - ref)
        print(name, 'ragged row', i, 'vs oracle rms err', e, 'rms ref', rms(ref))
        assert e < 1e-4 and e < 5e-4 * max(rms(ref), 1e-3), (name, i, e, rms(ref))


def test_ragged_64_clips_of_64_lengths_ctsnet():
    """64 clips of 64 distinct lengths in ONE call on the model where zero-padding is not neutral."""
    import torch
    rng = np.random.default_rng(7)
    lengths = sorted(set(int(v) for v in rng.integers(2400, 12000, 200)))[:64]
    rng.shuffle(lengths)
    assert len(set(lengths)) == 64
    clips, x = _ragged_batch(lengths, 2000)
    m = _make('ctsnet', 64, max(lengths), p_in=0.5, p_out=2.0)
    xt = torch.from_numpy(x).cuda()
    y = m.enhance_ragged(xt, lengths).cpu().numpy()
    one = _make('ctsnet', 1, max(lengths), p_in=0.5, p_out=2.0)
    worst = 0.0
    for i, c in enumerate(clips):
        ref = one.enhance_batch(torch.from_numpy(c[None]).cuda()).cpu().numpy()[0]
        e = rms(y[i, :len(ref)] - ref)
        worst = max(worst, e / max(rms(ref), 1e-3))
        assert e < 1e-4, (i, lengths[i], e)
    print('ctsnet ragged 64 x 64 lengths: worst relative rms err', worst)
    for i in (0, 31, 63):
        ref = _oracle('ctsnet', m, clips[i])
        assert rms(y[i, :len(ref)] - ref) < 1e-4
    # control: the same rows zero-padded to the longest clip and decoded as an equal-length batch are NOT the per-clip
    # results (that is why the lengths have to reach the norms) - the shortest clip moves by far more than the bar
    xz = x.copy()
    for i, n in enumerate(lengths):
        xz[i, n:] = 0.0
    yz = m.enhance_batch(torch.from_numpy(xz).cuda()).cpu().numpy()
    k = int(np.argmin(lengths))
    assert rms(yz[k, :lengths[k]] - y[k, :lengths[k]]) > 3e-4         # 3x the parity bar (measured 6.5e-4, 40 % of the signal)


def test_ragged_argument_checks():
    import torch
    from se_amd.engine import EngineError
    m = _make('crn', 2, 4000)
    x = torch.zeros((2, 4000), device='cuda') + 0.01
    with pytest.raises(EngineError):
        m.enhance_ragged(x, [4000, 100])            # shorter than one FFT frame
    with pytest.raises(EngineError):
        m.enhance_ragged(x, [4000, 5000])           # longer than the row
    with pytest.raises(EngineError):
        m.enhance_ragged(x, [4000])                 # one length per row


@pytest.mark.parametrize('name', ['ctsnet', 'g2net', 'taylorsenet', 'ctsnet_new', 'g2net_new', 'taylorsenet_new'])
def test_fused_tcm_block_matches_multi_launch_path(name):
    """From batch 96 a TCM / GLU block runs as ONE kernel per utterance (k_tcm.hip; round 3: also with the cumulative-LayerNorm
    heads of the `_new` variants); below, as 3-4 GEMM + 2-3 norm launches.  A batch of 100 (fused) must reproduce, clip for clip, what the fixture-pinned small-batch path gives - for
    equal lengths and for a ragged batch (statistics over each row's own frames inside the fused kernel)."""
    import torch
    L, B = 6000, 100
    clips = np.stack([synth.synth_clip(1500 + i, 'speech' if i % 3 else 'white', L) for i in range(5)])
    x = clips[np.arange(B) % 5].copy()
    big = _make(name, B, L, p_in=0.5, p_out=2.0)
    small = _make(name, 5, L, p_in=0.5, p_out=2.0)
    yb = big.enhance_batch(torch.from_numpy(x).cuda()).cpu().numpy()
    ys = small.enhance_batch(torch.from_numpy(clips).cuda()).cpu().numpy()
    for k in range(B):
        e = rms(yb[k] - ys[k % 5])
        assert e < 2e-5 * max(rms(ys[k % 5]), 1e-3), (name, k, e)
    lengths = [L - 97 * (k % 7) for k in range(B)]
    yr = big.enhance_ragged(torch.from_numpy(x).cuda(), lengths).cpu().numpy()
    one = _make(name, 1, L, p_in=0.5, p_out=2.0)
    for k in (0, 1, 6, 13, 99):
        n = lengths[k]
        ref = one.enhance_batch(torch.from_numpy(x[k:k + 1, :n].copy()).cuda()).cpu().numpy()[0]
        e = rms(yr[k, :len(ref)] - ref)
        assert e < 2e-5 * max(rms(ref), 1e-3), (name, 'ragged', k, e)
