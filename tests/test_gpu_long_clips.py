"""GPU: clips LONGER than BASELINE's 4 s against the reference itself.  VoiceBank+DEMAND reaches ~10-15 s; the fixtures
tests/golden/long10_<name>.npz (160 000 samples: T = 1001 / 1251 / 626) and long15_<name>.npz (239 987 samples, a multiple
of no hop: T = 1500 / 1876 / 938) hold the imported reference's decode of one clip each per network
(`python -m oracle.gen_golden --long`).  Paths only long clips reach: `tcm_fused_kernel`'s LDS tile (strip epilogue up to
T = 416, plain epilogue up to 512, the multi-launch fallback inside a B >= 96 call above), cumulative-LayerNorm scans and
ShareSepConv FIR history past 401 frames, dilation-128 halos, Uformer's attention over more keys than one LDS block holds
(Uformer/t_att_cplx.py:25 has no length limit - the engine streams the keys in blocks of 512), iSTFT windows.

Each fixture is decoded (a) alone and (b) as two rows of ONE ragged call among shorter clips - 96 rows for the TCM models
(the batch from which their blocks run fused), fewer for the others.  Bar: 1e-4 RMS on the waveform."""
import numpy as np
import pytest

import se_amd  # noqa: F401
from se_amd import synth
from conftest import load_golden, rms
from test_gpu_b256_fixture import SEEDS, make

pytestmark = pytest.mark.gpu
TCM = ('ctsnet', 'g2net', 'taylorsenet', 'ctsnet_new', 'g2net_new', 'taylorsenet_new')
ROWS = {**{n: 96 for n in TCM}, 'fullsubnet': 4, 'uformer': 12}


def _fixture(tag, name):
    G = load_golden(f'long{tag}_{name}')
    return synth.synth_clip(int(G['seed']), 'speech', int(G['n'])), G['enh_cprs']


@pytest.mark.parametrize('name', sorted(SEEDS))
def test_long_clips_match_reference(name):
    import torch
    c10, r10 = _fixture('10', name)
    c15, r15 = _fixture('15', name)
    Lmax = len(c15)
    # (a) alone
    one = make(name, 1, Lmax)
    for clip, ref, tag in ((c10, r10, '10 s'), (c15, r15, '15 s')):
        y = one.enhance_batch(torch.from_numpy(clip[None]).cuda()).cpu().numpy()[0]
        assert y.shape == ref.shape, (name, tag, y.shape, ref.shape)
        e = rms(y - ref)
        print(name, tag, 'alone: rms err vs reference', e, 'rms ref', rms(ref))
        assert np.isfinite(y).all() and e < 1e-4 and e < 5e-4 * max(rms(ref), 1e-3), (name, tag, e, rms(ref))
    del one
    # (b) rows of one ragged call
    B = ROWS.get(name, 16)
    rng = np.random.default_rng(11)
    lengths = [int(v) for v in rng.integers(16000, 150000, B)]
    r15_row, r10_row = 1, B - 2
    lengths[r15_row], lengths[r10_row] = len(c15), len(c10)
    x = np.zeros((B, Lmax), np.float32)
    short = synth.synth_clip(4242, 'speech', Lmax)
    for i, n in enumerate(lengths):
        x[i, :n] = short[:n] * (0.5 + 0.1 * (i % 7))
        x[i, n:] = 0.3                                   # junk past the end must be ignored
    x[r15_row, :len(c15)] = c15
    x[r10_row, :len(c10)] = c10
    big = make(name, B, Lmax)
    y = big.enhance_ragged(torch.from_numpy(x).cuda(), lengths).cpu().numpy()
    assert np.isfinite(y).all()
    for row, ref, tag in ((r15_row, r15, '15 s'), (r10_row, r10, '10 s')):
        e = rms(y[row, :len(ref)] - ref)
        print(name, tag, f'row {row} of a ragged batch of {B}: rms err vs reference', e)
        assert e < 1e-4 and e < 5e-4 * max(rms(ref), 1e-3), (name, tag, 'ragged', e, rms(ref))
        assert not y[row, len(ref):].any()
