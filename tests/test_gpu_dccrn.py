"""GPU parity: the HIP engine (through the C ABI) vs the numpy oracle and the reference-generated fixtures."""
import numpy as np
import pytest

import se_amd
from se_amd import synth
from conftest import load_golden, rms

pytestmark = pytest.mark.gpu

CTOR = dict(rnn_units=256, masking_mode='E', use_clstm=True, kernel_num=[32, 64, 128, 256, 256, 256])


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


def test_stft_istft_stage_hooks():
    torch = _torch()
    from se_amd.models import DCCRN
    from oracle import stft as S
    eng = DCCRN(**CTOR).load_synthetic(14).engine
    x = synth.synth_batch(3, 'speech', 4096, seed0=20)          # multiple of hop -> no tail pad
    xd = torch.from_numpy(x).cuda()
    spec = eng.stft(xd).cpu().numpy()                            # [B,2,F,T]
    ref = S.stft(x.astype(np.float64), 512, 128)
    assert spec.shape == (3, 2, 257, ref.shape[-1])
    err = rms((spec[:, 0] + 1j * spec[:, 1]) - ref)
    assert err < 2e-6 * rms(np.abs(ref)), err
    # power compression
    spec_c = eng.stft(xd, p_in=0.5).cpu().numpy()
    refc = np.abs(ref) ** 0.5 * np.exp(1j * np.angle(ref))
    assert rms((spec_c[:, 0] + 1j * spec_c[:, 1]) - refc) < 5e-6 * rms(np.abs(refc))
    # inverse round trip (size-independent property)
    y = eng.istft(torch.from_numpy(np.ascontiguousarray(spec)).cuda(), 4096).cpu().numpy()
    assert rms(y - x) < 2e-6 * rms(x)
    c = eng.rms_scale(xd).cpu().numpy()
    assert np.allclose(c, S.rms_scale(x), rtol=1e-6)


def test_forward_matches_reference_fixture():
    torch = _torch()
    from se_amd.models import DCCRN
    G = load_golden('dccrn')
    m = DCCRN(**CTOR).load_synthetic(14)
    y = m(torch.from_numpy(G['x']).cuda()).cpu().numpy()
    assert y.shape == G['y'].shape
    err = rms(y - G['y'])
    assert err < 1e-5 * max(rms(G['y']), 1.0), (err, rms(G['y']))


@pytest.mark.parametrize('mode', ['C', 'R'])
def test_masking_modes_match_reference_fixture(mode):
    """DCCRN(masking_mode='C' | 'R') (DCCRN/DCCRN_cprs.py:220-223; SE_CFG_DCCRN_MASK_C / _R): forward and compressed-spectrum
    decode against the imported reference's output, the decode streamed as well (the mask is per frame)."""
    torch = _torch()
    from se_amd.models import DCCRN
    G = load_golden('dccrn_mask')
    m = DCCRN(**dict(CTOR, masking_mode=mode), p_in=0.5, p_out=2.0, max_batch=2).load_synthetic(14)
    y = m(torch.from_numpy(G['x']).cuda()).cpu().numpy()
    err = rms(y - G['y_' + mode])
    assert y.shape == G['y_' + mode].shape and err < 1e-5 * max(rms(G['y_' + mode]), 1.0), (err, rms(G['y_' + mode]))
    wav = torch.from_numpy(np.stack([G['wav'], G['wav'][::-1].copy()])).cuda()
    out = m.enhance_batch(wav).cpu().numpy()
    ref = G['enh_cprs_' + mode]
    e = rms(out[0] - ref)
    print('masking mode', mode, 'decode rms err', e, 'rms ref', rms(ref))
    assert out.shape[1] == ref.shape[0] and e < 1e-4 and e < 5e-4 * max(rms(ref), 1e-3)
    assert rms(out[0] - DCCRN(**CTOR, p_in=0.5, p_out=2.0, max_batch=2).load_synthetic(14).enhance_batch(wav).cpu().numpy()[0]) > 1e-2 * rms(ref)
    eng = m.engine
    eng.stream_begin(2, c=eng.rms_scale(wav), max_chunk_frames=4)
    outs = [eng.stream_push(wav[:, :1700].contiguous()).cpu().numpy(), eng.stream_push(wav[:, 1700:].contiguous()).cpu().numpy(),
            eng.stream_flush().cpu().numpy()]
    got = np.concatenate(outs, axis=1)
    assert got.shape == out.shape and rms(got - out) < 1e-6 + 2e-5 * rms(out)


@pytest.mark.parametrize('p_in,p_out,key', [(1.0, 1.0, 'enh'), (0.5, 2.0, 'enh_cprs')])
def test_enhance_matches_reference_fixture(p_in, p_out, key):
    torch = _torch()
    from se_amd.models import DCCRN
    G = load_golden('dccrn')
    m = DCCRN(**CTOR, p_in=p_in, p_out=p_out, max_batch=2).load_synthetic(14)
    wav = torch.from_numpy(np.stack([G['wav'], G['wav']])).cuda()
    y = m.enhance_batch(wav).cpu().numpy()
    assert y.shape[1] == G[key].shape[0]
    for b in range(2):
        err = rms(y[b] - G[key])
        assert err < 1e-4, err                       # north_star: <= 1e-4 RMS on the waveform
        assert err < 5e-4 * max(rms(G[key]), 1e-3), (err, rms(G[key]))


def test_enhance_vs_oracle_batch_and_lengths():
    """Batched decode == per-utterance oracle decode; ragged length (not a hop multiple) exercises the tail pad."""
    torch = _torch()
    from se_amd.models import DCCRN
    from oracle import decode as D
    m = DCCRN(**CTOR, p_in=0.5, p_out=2.0, max_batch=3, max_samples=6000).load_synthetic(14)
    sd = synth.synth_state_dict(m.state_dict_schema(), 14)
    for L, kinds in ((5000, ('speech', 'white', 'gap')), (1500, ('quiet', 'speech', 'white'))):
        x = np.stack([synth.synth_clip(30 + i, k, L) for i, k in enumerate(kinds)])
        y = m.enhance_batch(torch.from_numpy(x).cuda()).cpu().numpy()
        for b in range(3):
            ref = D.enhance_dccrn(sd, x[b], 0.5, 2.0)
            assert y[b].shape == ref.shape
            e = rms(y[b] - ref)
            print(L, kinds[b], "rms err", e, "rms ref", rms(ref))
            assert e < 1e-4 and e < 5e-4 * max(rms(ref), 1e-3), (L, kinds[b], e, rms(ref))


@pytest.mark.parametrize('flags', [2, 4, 6])
def test_complexnn_convention_flags(flags):
    """SE_CFG_DCCRN_BIAS_PER_PART (2) / SE_CFG_DCCRN_PLAIN_CAT (4): the conventions of the absent `complexnn` that
    DCCRN_cprs.py does not determine are weight-preparation switches; each matches the oracle's same variant and differs
    from the default (so the switch is live)."""
    torch = _torch()
    from se_amd.models import DCCRN
    from oracle import models as M
    G = load_golden('dccrn')
    sd = synth.synth_state_dict(DCCRN.state_dict_schema(), 14)
    x = torch.from_numpy(G['x']).cuda()
    y = DCCRN(**CTOR, flags=flags).load_synthetic(14)(x).cpu().numpy()
    ref = M.dccrn_forward(sd, G['x'], variant=flags)
    assert rms(y - ref) < 1e-5 * max(rms(ref), 1.0)
    assert rms(y - G['y']) > 1e-3 * rms(G['y'])
