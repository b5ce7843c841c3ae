"""GPU: the se_stft / se_istft / se_rms_scale stage hooks diffed ALONE (not through a network) at every front-end geometry
of the zoo, against oracle/stft.py (itself pinned to torch.stft / torch.istft fp64 fixtures,
tests/test_oracle_golden.py) and against the torch fixtures directly:
  320/160/320  librosa family + CTSNet / TaylorSENet    (engine of CRN)
  512/128/512  DCCRN
  512/256/512  FullSubNet
  512/160/400  Uformer (window shorter than the FFT, centred)
Lengths: the fixture's 4 000 samples, BASELINE's 64 000, and a ragged 5 003 (not a hop multiple)."""
import numpy as np
import pytest

import se_amd  # noqa: F401
from se_amd import synth
from conftest import load_golden, rms

pytestmark = pytest.mark.gpu
GEOMS = [('crn', 320, 160, 320), ('dccrn', 512, 128, 512), ('fullsubnet', 512, 256, 512), ('uformer', 512, 160, 400)]
WSEED = {'crn': 12, 'dccrn': 14, 'fullsubnet': 15, 'uformer': 21}


def _engine(name, B, L):
    from se_amd.models import MODEL_CLASSES
    return MODEL_CLASSES[name](max_batch=B, max_samples=L).load_synthetic(WSEED[name]).engine


@pytest.mark.parametrize('name,n_fft,hop,win', GEOMS)
def test_stft_istft_hooks_match_torch_fixture(name, n_fft, hop, win):
    import torch
    G = load_golden('stft')
    tag = f'{n_fft}_{hop}_{win}'
    x = G['x_' + tag].astype(np.float32)                               # [2, 4000]
    eng = _engine(name, 2, 8000)
    spec = eng.stft(torch.from_numpy(x).cuda()).cpu().numpy()
    ref = G['spec_' + tag]                                             # torch.stft fp64, [2, F, T]
    if name == 'dccrn':                                                # DCCRN's script tail-pads to a hop multiple first
        T = ref.shape[-1]
        spec = spec[..., :T]                                           # frames that do not touch the padding are equal
        ok = slice(0, T - 3)
    else:
        ok = slice(None)
    assert spec.shape[2] == ref.shape[1]
    got = spec[:, 0] + 1j * spec[:, 1]
    assert got.shape[-1] == ref.shape[-1] or name == 'dccrn'
    e = rms((got - ref)[..., ok])
    assert e < 2e-6 * rms(np.abs(ref)), (name, e)
    # inverse: the fp64 torch spectrum through se_istft == torch.istft(length=4000)
    sp = np.ascontiguousarray(np.stack([ref.real, ref.imag], axis=1).astype(np.float32))
    y = eng.istft(torch.from_numpy(sp).cuda(), 4000).cpu().numpy()
    assert rms(y - G['ylen_' + tag]) < 2e-6 * rms(G['ylen_' + tag])
    # no `length` (CTSNet `[:wav_len]`, Uformer): hop * (T - 1) samples
    n2 = G['ynolen_' + tag].shape[-1]
    y2 = eng.istft(torch.from_numpy(sp).cuda(), n2).cpu().numpy()
    assert rms(y2 - G['ynolen_' + tag]) < 2e-6 * rms(G['ynolen_' + tag])


@pytest.mark.parametrize('name,n_fft,hop,win', GEOMS)
@pytest.mark.parametrize('L', [64000, 5003])
def test_stft_istft_hooks_match_oracle_at_size(name, n_fft, hop, win, L):
    import torch
    from oracle import stft as S
    B = 3
    x = np.stack([synth.synth_clip(700 + i, k, L) for i, k in enumerate(('speech', 'white', 'gap'))])
    eng = _engine(name, B, L)
    xd = torch.from_numpy(x).cuda()
    c = eng.rms_scale(xd)
    cr = S.rms_scale(x)
    assert np.allclose(c.cpu().numpy(), cr, rtol=2e-6)
    spec = eng.stft(xd, c=c, p_in=0.5).cpu().numpy()
    T = eng.num_frames(L)
    assert spec.shape == (B, 2, n_fft // 2 + 1, T)
    xs = x.astype(np.float64) * cr[:, None]
    if name == 'dccrn':
        xs = S.pad_to_hop(xs, n_fft, hop)
    ref = S.stft(xs, n_fft, hop, win)
    assert ref.shape[-1] == T
    refc = np.abs(ref) ** 0.5 * np.exp(1j * np.angle(ref))
    got = spec[:, 0] + 1j * spec[:, 1]
    e = rms(got - refc)
    # sqrt compression doubles the relative weight of the small bins, where the fp32 FFT's absolute error (set by the
    # frame's energy) is largest relative to the bin: 1.5e-5 compressed, 3e-6 uncompressed
    assert e < 1.5e-5 * rms(np.abs(refc)), (name, L, e)
    spec1 = eng.stft(xd, c=c).cpu().numpy()
    e1 = rms((spec1[:, 0] + 1j * spec1[:, 1]) - ref)
    assert e1 < 3e-6 * rms(np.abs(ref)), (name, L, e1)
    # inverse with the de-normalisation: se_istft(spec, c) == istft(spec) / c
    n_out = int(eng.output_samples(L)) if name != 'dccrn' else xs.shape[-1]
    sp = np.ascontiguousarray(np.stack([ref.real, ref.imag], axis=1).astype(np.float32))
    y = eng.istft(torch.from_numpy(sp).cuda(), n_out, c=c).cpu().numpy()
    yr = S.istft(ref, n_fft, hop, win, length=n_out) / cr[:, None]
    assert rms(y - yr) < 3e-6 * rms(yr), (name, L, rms(y - yr), rms(yr))


# ---- se_frontend / se_backend: the two halves of a decode loop's body around `model(feat)`, each diffed alone
# against the lines of oracle/decode.py that restate them (a1-a5 and a12 / a17-a20).  p_in / p_out come from the engine.
def _engine_p(name, B, L, p_in, p_out):
    from se_amd.models import MODEL_CLASSES
    return MODEL_CLASSES[name](max_batch=B, max_samples=L, p_in=p_in, p_out=p_out).load_synthetic(WSEED[name]).engine


@pytest.mark.parametrize('name,n_fft,hop,win', [g for g in GEOMS if g[0] != 'uformer'])
@pytest.mark.parametrize('p_in,p_out', [(1.0, 1.0), (0.5, 2.0)])
def test_frontend_and_backend_hooks_match_oracle(name, n_fft, hop, win, p_in, p_out):
    import torch
    from oracle import stft as S
    B, L = 3, 16000 if name != 'dccrn' else 16013           # (DCCRN: a length the script has to tail-pad)
    x = np.stack([synth.synth_clip(900 + i, k, L) for i, k in enumerate(('speech', 'white', 'gap'))])
    eng = _engine_p(name, B, L, p_in, p_out)
    xd = torch.from_numpy(x).cuda()
    # front end: c, and the compressed spectrum the network is fed (e.g. dccrn_decode_vb.py:26-42)
    c, spec = eng.frontend(xd)
    cr = S.rms_scale(x)
    assert np.allclose(c.cpu().numpy(), cr, rtol=2e-6)
    xs = x.astype(np.float64) * cr[:, None]
    if name == 'dccrn':
        xs = S.pad_to_hop(xs, n_fft, hop)
    ref = S.stft(xs, n_fft, hop, win)
    feat = np.abs(ref) ** p_in * np.exp(1j * np.angle(ref))
    T, F = eng.num_frames(L), n_fft // 2 + 1
    sp = spec.cpu().numpy()
    assert sp.shape == (B, 2, F, T) and ref.shape[-1] == T
    e = rms((sp[:, 0] + 1j * sp[:, 1]) - feat)
    assert e < 1.5e-5 * rms(np.abs(feat)), (name, 'frontend', e)
    n_out = int(eng.output_samples(L))
    rng = np.random.default_rng(3)
    spec32 = np.ascontiguousarray(np.stack([feat.real, feat.imag], 1).astype(np.float32))
    sd = torch.from_numpy(spec32).cuda()
    featc = spec32[:, 0].astype(np.float64) + 1j * spec32[:, 1].astype(np.float64)

    def inv(de):
        return S.istft(de, n_fft, hop, win, length=n_out) / cr[:, None]

    # back end, complex-mapping scripts: the estimate's own magnitude / phase (gcrn_decode_vb.py:47-58)
    est = np.ascontiguousarray((rng.standard_normal((B, 2, F, T)) * np.abs(feat)[:, None]).astype(np.float32))
    y = eng.backend('ri', torch.from_numpy(est).cuda(), n_out, c=c).cpu().numpy()
    ec = est[:, 0].astype(np.float64) + 1j * est[:, 1].astype(np.float64)
    yr = inv(np.abs(ec) ** p_out * np.exp(1j * np.angle(ec)))
    assert rms(y - yr) < 5e-6 * rms(yr), (name, 'ri', rms(y - yr), rms(yr))
    # magnitude mapping + noisy phase (lstm_decode_vb.py:47-52)
    mag = np.ascontiguousarray((np.abs(feat) * rng.uniform(0.1, 1.0, feat.shape)).astype(np.float32))
    y = eng.backend('mag', torch.from_numpy(mag).cuda(), n_out, spec=sd, c=c).cpu().numpy()
    yr = inv(mag.astype(np.float64) ** p_out * np.exp(1j * np.angle(featc)))
    assert rms(y - yr) < 5e-6 * rms(yr), (name, 'mag', rms(y - yr), rms(yr))
    # complex ratio mask on the noisy spectrum (fullsubnet_sa_decode_vb.py:56-72)
    mask = rng.standard_normal((B, 2, F, T)).astype(np.float32)
    y = eng.backend('cmask', torch.from_numpy(mask).cuda(), n_out, spec=sd, c=c).cpu().numpy()
    mc = mask[:, 0].astype(np.float64) + 1j * mask[:, 1].astype(np.float64)
    es = mc * featc
    yr = inv(np.abs(es) ** p_out * np.exp(1j * np.angle(es)))
    assert rms(y - yr) < 5e-6 * rms(yr), (name, 'cmask', rms(y - yr), rms(yr))
