"""GPU: every network of SURVEY 8(a) decoded at BASELINE's FULL clip size (16 kHz x 4 s = 64 000 samples, T = 401 / 501 /
251) against a fixture produced by the imported reference (oracle/gen_golden.py save_full, `--full`): a T-tiling bug
(attention key chunks, cLN prefix scan, dilated-TCM halo at dilation > T of the small fixtures) cannot hide behind the
4 000-sample fixtures.  Compressed exponents 0.5 / 2.0 everywhere.  Bar: 1e-4 RMS on the waveform (north star)."""
import numpy as np
import pytest

import se_amd  # noqa: F401
from se_amd import synth
from conftest import load_golden, rms

pytestmark = pytest.mark.gpu
L = 64000
# name -> synthetic-weight seed(s) of the fixture
SEEDS = {'lstm': 11, 'crn': 12, 'fullsubnet': 15, 'gcrn': 16, 'taylorsenet': 19, 'g2net': 20, 'uformer': 21,
         'taylorsenet_new': 19, 'g2net_new': 20, 'ctsnet': (17, 18), 'ctsnet_new': (17, 18)}


def _model(name, max_batch):
    from se_amd import models, models_new  # noqa: F401
    kw = dict(max_batch=max_batch, max_samples=L, p_in=0.5, p_out=2.0)
    if name == 'uformer':
        kw = dict(max_batch=max_batch, max_samples=L)              # in-model STFT; no exponents in its script
    if name.startswith('ctsnet'):
        cls = models_new.CTSNet if name.endswith('_new') else models.CTSNet
        return cls(**kw).load_synthetic(*SEEDS[name])
    return models.MODEL_CLASSES[name](**kw).load_synthetic(SEEDS[name])


@pytest.mark.parametrize('name', sorted(SEEDS))
def test_full_size_decode_matches_reference_fixture(name):
    import torch
    G = load_golden('full_' + name)
    assert int(G['n']) == L
    wav = synth.synth_clip(int(G['seed']), 'speech', L)
    # the fixture clip rides in a batch of 3 next to other clips (row 1), so batching is exercised at full size too
    x = np.stack([synth.synth_clip(900, 'white', L), wav, synth.synth_clip(901, 'speech', L)])
    m = _model(name, 3)
    y = m.enhance_batch(torch.from_numpy(x).cuda()).cpu().numpy()
    ref = G['enh4_cprs']
    assert y.shape[1] == ref.shape[0], (y.shape, ref.shape)
    e = rms(y[1] - ref)
    print(name, 'full-size decode rms err', e, 'rms ref', rms(ref))
    assert np.isfinite(y).all()
    assert e < 1e-4 and e < 5e-4 * max(rms(ref), 1e-3), (name, e, rms(ref))


def test_dccrn_full_size_decode_matches_reference_fixture():
    """tests/golden/dccrn.npz:enh4_cprs - the imported DCCRN_cprs.py (on oracle/_complexnn_recall.py) on a 4 s clip."""
    import torch
    from se_amd.models import MODEL_CLASSES
    G = load_golden('dccrn')
    wav = synth.synth_clip(1, 'speech', L)
    x = np.stack([wav, synth.synth_clip(902, 'white', L)])
    m = MODEL_CLASSES['dccrn'](max_batch=2, max_samples=L, p_in=0.5, p_out=2.0).load_synthetic(14)
    y = m.enhance_batch(torch.from_numpy(x).cuda()).cpu().numpy()
    ref = G['enh4_cprs']
    assert y.shape[1] == ref.shape[0]
    e = rms(y[0] - ref)
    print('dccrn full-size decode rms err', e, 'rms ref', rms(ref))
    assert e < 1e-4 and e < 5e-4 * max(rms(ref), 1e-3)
