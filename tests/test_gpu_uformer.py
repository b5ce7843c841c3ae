"""GPU parity for Uformer (STFT / iSTFT inside the model) vs the reference-generated fixture and the oracle."""
import numpy as np
import pytest

import se_amd
from se_amd import synth
from conftest import load_golden, rms

pytestmark = pytest.mark.gpu


def test_uformer_decode_matches_reference():
    import torch
    from se_amd.models import Uformer
    from oracle import decode as D
    G = load_golden('uformer')
    m = Uformer(max_batch=2, max_samples=4000).load_synthetic(21)
    wav2 = synth.synth_clip(55, 'white', 4000)
    y = m.enhance_batch(torch.from_numpy(np.stack([G['wav'], wav2])).cuda()).cpu().numpy()
    e = rms(y[0] - G['enh'])
    print('uformer decode rms err', e, rms(G['enh']))
    assert y.shape[1] == G['enh'].shape[0]
    assert e < 1e-4 and e < 5e-4 * max(rms(G['enh']), 1e-3)
    sd = synth.synth_state_dict(m.state_dict_schema(), 21)
    ref2 = D.enhance_uformer(sd, wav2)
    e2 = rms(y[1] - ref2)
    print('uformer decode (oracle, white) rms err', e2, rms(ref2))
    assert e2 < 1e-4 and e2 < 5e-4 * max(rms(ref2), 1e-3)
    # forward(wav, wav)[0] is the un-normalised path: model output for the unit-RMS input
    c = float(np.sqrt(len(G['wav']) / np.sum(G['wav'].astype(np.float64) ** 2)))
    out = m(torch.from_numpy((G['wav'].astype(np.float64) * c).astype(np.float32)[None]).cuda())[0].cpu().numpy()[0]
    assert rms(out / c - G['enh']) < 1e-4


def test_uformer_full_return_matches_reference():
    """`output, src, output_cplx, src_cplx = model(inputs, src)` (uformer.py:287) against the reference's own 4-tuple
    (tests/golden/uformer.npz: cplx, src_wav, src_cplx) with a source that differs from the input; in a batch of 2."""
    import torch
    from se_amd.models import Uformer
    G = load_golden('uformer')
    c = float(np.sqrt(len(G['wav']) / np.sum(G['wav'].astype(np.float64) ** 2)))
    x = (G['wav'].astype(np.float64) * c).astype(np.float32)
    src = (synth.synth_clip(13, 'speech', 4000).astype(np.float64) * c).astype(np.float32)
    other = synth.synth_clip(56, 'white', 4000)
    m = Uformer(max_batch=2, max_samples=4000).load_synthetic(21)
    out, src_out, out_c, src_c = m(torch.from_numpy(np.stack([other, x])).cuda(), torch.from_numpy(np.stack([x, src])).cuda())
    assert out_c.shape == (2, 2, 257, 26) and src_c.shape == (2, 2, 257, 26) and src_out.shape == out.shape == (2, 4000)
    for name, got, ref in (('output', out[1].cpu().numpy() / c, G['enh']), ('src', src_out[1].cpu().numpy(), G['src_wav'][0]),
                           ('output_cplx', out_c[1].cpu().numpy(), G['cplx'][0]),
                           ('src_cplx', src_c[1].cpu().numpy(), G['src_cplx'][0])):
        e = rms(got - ref)
        print('uformer 4-tuple', name, 'rms err', e, 'rms ref', rms(ref))
        assert got.shape == ref.shape and e < 1e-4 and e < 5e-4 * max(rms(ref), 1e-3), (name, e, rms(ref))
    # without a source the source outputs are None; spectra=False skips the RI tensors
    o2, s2, c2, sc2 = m(torch.from_numpy(x[None]).cuda(), spectra=False)
    assert s2 is None and c2 is None and sc2 is None and torch.equal(o2[0], m(torch.from_numpy(x[None]).cuda())[0][0])
