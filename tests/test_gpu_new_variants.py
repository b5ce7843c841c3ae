"""GPU parity for the `*_new` directories (CTSNet_new, TaylorSENet_new, G2Net_new): the base networks with every
InstanceNorm replaced by CumulativeLayerNorm, decoded with the compressed 0.5 / 2.0 exponents.  Fixtures come from
the imported reference modules (oracle/gen_golden.py gen_*_new)."""
import numpy as np
import pytest

import se_amd
from se_amd import synth
from conftest import load_golden, rms

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize('name,seed', [('taylorsenet_new', 19), ('g2net_new', 20)])
def test_forward_and_decode_match_reference(name, seed):
    torch = _torch()
    from se_amd import models_new  # noqa: F401  (registers the *_new builders)
    from se_amd.models import MODEL_CLASSES
    G = load_golden(name)
    m = MODEL_CLASSES[name](max_batch=2, max_samples=8000).load_synthetic(seed)
    assert (m.p_in, m.p_out) == (0.5, 2.0)
    y = m(torch.from_numpy(G['x']).cuda())
    y = (y[-1] if isinstance(y, list) else y).cpu().numpy()
    err = rms(y - G['y'])
    print(name, 'forward rms err', err, 'rms ref', rms(G['y']))
    assert y.shape == G['y'].shape and err < 2e-5 * max(rms(G['y']), 1.0)
    wav = torch.from_numpy(np.stack([G['wav'], G['wav'][::-1].copy()])).cuda()
    e = m.enhance_batch(wav).cpu().numpy()
    err = rms(e[0] - G['enh_cprs'])
    print(name, 'decode rms err', err, 'rms ref', rms(G['enh_cprs']))
    assert err < 1e-4 and err < 5e-4 * max(rms(G['enh_cprs']), 1e-3)


def test_ctsnet_new_matches_reference():
    torch = _torch()
    from se_amd.models_new import Step1_net, Step2_net, CTSNet
    G = load_golden('ctsnet_new')
    m1 = Step1_net(max_batch=2, max_samples=8000).load_synthetic(17)
    e1 = rms(m1(torch.from_numpy(G['x1']).cuda()).cpu().numpy() - G['y1'])
    m2 = Step2_net(X=6, R=3, max_batch=2, max_samples=8000).load_synthetic(18)
    e2 = rms(m2(torch.from_numpy(G['x2']).cuda()).cpu().numpy() - G['y2'])
    print('cts_new stage rms err', e1, e2)
    assert e1 < 2e-5 * max(rms(G['y1']), 1.0) and e2 < 2e-5 * max(rms(G['y2']), 1.0)
    net = CTSNet(max_batch=2, max_samples=8000).load_synthetic(17, 18)
    wav = torch.from_numpy(np.stack([G['wav'], synth.synth_clip(77, 'white', 8000)])).cuda()
    y = net.enhance_batch(wav).cpu().numpy()
    e = rms(y[0] - G['enh_cprs'])
    print('ctsnet_new decode rms err', e, rms(G['enh_cprs']))
    assert e < 1e-4 and e < 5e-4 * max(rms(G['enh_cprs']), 1e-3)


def test_mixed_norm_keys_are_rejected():
    """A state dict with an InstanceNorm key where the cLN variant expects `gain` must fail the strict load."""
    torch = _torch()
    from se_amd.engine import Engine
    from se_amd import schemas
    sd = synth.synth_state_dict(schemas.SCHEMAS['taylorsenet_new'](), 1)
    k = next(k for k in sd if k.endswith('.gain'))
    sd[k[:-4] + 'weight'] = sd[k].reshape(-1)
    eng = Engine('taylorsenet', 0, 1, 4000, 0.5, 2.0)
    with pytest.raises(RuntimeError):
        eng.load_state_dict(sd)
