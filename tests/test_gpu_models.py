"""GPU parity for LSTM / CRN / DPCRN: engine (through the C ABI) vs reference-generated fixtures and the numpy oracle.
DPCRN additionally runs with the reference's REAL checkpoints (fixture copies of DPCRN/BEST_MODEL/vb_dpcrn_*.pth)."""
import numpy as np
import pytest

import se_amd
from se_amd import synth
from conftest import load_golden, rms

pytestmark = pytest.mark.gpu
SEEDS = {'lstm': 11, 'crn': 12, 'dpcrn': 13, 'fullsubnet': 15, 'gcrn': 16, 'taylorsenet': 19, 'g2net': 20}


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


@pytest.mark.parametrize('name', ['lstm', 'crn', 'dpcrn', 'fullsubnet', 'gcrn', 'taylorsenet', 'g2net'])
def test_forward_matches_reference_fixture(name):
    torch = _torch()
    from se_amd.models import MODEL_CLASSES
    G = load_golden(name)
    m = MODEL_CLASSES[name](max_batch=2, max_samples=8000).load_synthetic(SEEDS[name])
    y = m(torch.from_numpy(G['x']).cuda())
    y = (y[-1] if isinstance(y, list) else y).cpu().numpy()
    assert y.shape == G['y'].shape
    err = rms(y - G['y'])
    print(name, 'forward rms err', err, 'rms ref', rms(G['y']))
    assert err < 2e-5 * max(rms(G['y']), 1.0), (err, rms(G['y']))


@pytest.mark.parametrize('name', ['lstm', 'crn', 'dpcrn', 'fullsubnet', 'gcrn', 'taylorsenet', 'g2net'])
def test_enhance_matches_reference_fixture(name):
    torch = _torch()
    from se_amd.models import MODEL_CLASSES
    G = load_golden(name)
    m = MODEL_CLASSES[name](max_batch=2, max_samples=8000).load_synthetic(SEEDS[name])
    wav = torch.from_numpy(np.stack([G['wav'], G['wav'][::-1].copy()])).cuda()
    y = m.enhance_batch(wav).cpu().numpy()
    err = rms(y[0] - G['enh'])
    print(name, 'enhance rms err', err, 'rms ref', rms(G['enh']))
    assert y.shape[1] == G['enh'].shape[0]
    assert err < 1e-4 and err < 5e-4 * max(rms(G['enh']), 1e-3), (err, rms(G['enh']))


def test_fullsubnet_gru_sequence_model():
    """The GRU time step (`Model(sequence_model="GRU")`, sequence_model.py:36-43; SE_CFG_FSN_GRU): forward and decode
    against the reference-generated fixture, a ragged pair against the oracle, and the key schema is the GRU's."""
    torch = _torch()
    from se_amd.models import Model
    from oracle import decode as D
    G = load_golden('fullsubnet_gru')
    kw = dict(sb_num_neighbors=15, fb_num_neighbors=0, num_freqs=257, look_ahead=2, sequence_model="GRU",
              fb_output_activate_function="ReLU", sb_output_activate_function=None, fb_model_hidden_size=512,
              sb_model_hidden_size=384, weight_init=True, norm_type="offline_laplace_norm", num_groups_in_drop_band=2)
    m = Model(max_batch=2, max_samples=8000, p_in=0.5, p_out=2.0, **kw)
    assert m.state_dict_schema()['fb_model.sequence_model.weight_ih_l0'][0] == (3 * 512, 257)
    m.load_synthetic(25)
    y = m(torch.from_numpy(G['x']).cuda()).cpu().numpy()
    err = rms(y - G['y'])
    print('fullsubnet GRU forward rms err', err, 'rms ref', rms(G['y']))
    assert y.shape == G['y'].shape and err < 2e-5 * max(rms(G['y']), 1.0)
    other = synth.synth_clip(61, 'white', 5000)
    x = np.zeros((2, 6000), np.float32)
    x[0], x[1, :5000] = G['wav'], other
    out = m.enhance_ragged(torch.from_numpy(x).cuda(), [6000, 5000]).cpu().numpy()
    err = rms(out[0] - G['enh_cprs'])
    print('fullsubnet GRU decode rms err', err, 'rms ref', rms(G['enh_cprs']))
    assert err < 1e-4 and err < 5e-4 * max(rms(G['enh_cprs']), 1e-3)
    sd = synth.synth_state_dict(m.state_dict_schema(), 25)
    ref = D.ENHANCE['fullsubnet'](sd, other, 0.5, 2.0)
    assert rms(out[1, :5000] - ref) < 1e-4
    with pytest.raises(RuntimeError):                       # an LSTM state dict is not a GRU state dict (strict load)
        Model(max_batch=1, max_samples=8000, **kw).load_state_dict(synth.synth_state_dict(Model().state_dict_schema(), 15))


FSN_KW = dict(sb_num_neighbors=15, fb_num_neighbors=0, num_freqs=257, look_ahead=2, sequence_model="LSTM",
              fb_output_activate_function="ReLU", sb_output_activate_function=None, fb_model_hidden_size=512,
              sb_model_hidden_size=384, weight_init=True, num_groups_in_drop_band=2)


def test_fullsubnet_cumulative_norm_and_frame_online():
    """`Model(norm_type="cumulative_laplace_norm")` (base_model.py:212-240, SE_CFG_FSN_CUMULATIVE): forward and decode against
    the reference-generated fixture, a ragged pair against the oracle, and the frame-online mode (look_ahead = 2 frames: the
    engine finalises its estimate two frames late) - a 2 s clip pushed in pieces equals the offline decode, which the fixture
    pins to the reference's own output; the decode script's offline norm stays un-streamable."""
    torch = _torch()
    from se_amd.models import Model
    from oracle import decode as D
    G = load_golden('fullsubnet_cum')
    m = Model(max_batch=2, max_samples=32000, p_in=0.5, p_out=2.0, norm_type="cumulative_laplace_norm", **FSN_KW).load_synthetic(15)
    y = m(torch.from_numpy(G['x']).cuda()).cpu().numpy()
    err = rms(y - G['y'])
    print('fullsubnet cumulative forward rms err', err, 'rms ref', rms(G['y']))
    assert y.shape == G['y'].shape and err < 2e-5 * max(rms(G['y']), 1.0)
    other = synth.synth_clip(62, 'white', 5000)
    x = np.zeros((2, 6000), np.float32)
    x[0], x[1, :5000] = G['wav'], other
    out = m.enhance_ragged(torch.from_numpy(x).cuda(), [6000, 5000]).cpu().numpy()
    err = rms(out[0] - G['enh_cprs'])
    print('fullsubnet cumulative decode rms err', err, 'rms ref', rms(G['enh_cprs']))
    assert err < 1e-4 and err < 5e-4 * max(rms(G['enh_cprs']), 1e-3)
    sd = synth.synth_state_dict(m.state_dict_schema(), 15)
    ref = D.ENHANCE['fullsubnet'](sd, other, 0.5, 2.0, norm_type='cumulative_laplace_norm')
    assert rms(out[1, :5000] - ref) < 1e-4 and rms(out[1, :5000] - ref) < 5e-4 * max(rms(ref), 1e-3)
    # ---- frame-online: two streams (the fixture's 2 s clip and another), pieces of 10 ms ... 0.4 s, chunks of up to 5 frames
    L = 32000
    xs = np.stack([G['wav2'], synth.synth_clip(63, 'speech', L)])
    xt = torch.from_numpy(xs).cuda()
    off = m.enhance_batch(xt).cpu().numpy()
    e0 = rms(off[0] - G['enh2_cprs'])
    assert e0 < 1e-4 and e0 < 5e-4 * max(rms(G['enh2_cprs']), 1e-3), e0
    eng = m.engine
    for pieces, chunk in (([160, 37, 3000, 7, 6400], 5), ([4000], 16)):
        eng.stream_begin(2, c=eng.rms_scale(xt), max_chunk_frames=chunk)
        outs, pos, fed, emitted = [], 0, 0, 0
        k = 0
        while pos < L:
            n = min(pieces[min(k, len(pieces) - 1)], L - pos)
            outs.append(eng.stream_push(xt[:, pos:pos + n].contiguous()).cpu().numpy())
            pos += n
            k += 1
            fed, emitted = pos, emitted + outs[-1].shape[1]
            # output arrives within half a window + (look-ahead + 2) hops of the input
            assert emitted >= fed - 257 - 4 * 256 or fed < 512, (fed, emitted)
        outs.append(eng.stream_flush().cpu().numpy())
        got = np.concatenate(outs, axis=1)
        assert got.shape == off.shape, (got.shape, off.shape)
        e = rms(got - off)
        print('fullsubnet cumulative streamed vs offline rms err', e, 'rms ref', rms(off))
        assert e < 1e-6 + 2e-5 * rms(off), (e, rms(off))
    with pytest.raises(RuntimeError):                       # the decode script's utterance-mean norm is not causal
        Model(max_batch=1, max_samples=8000, norm_type="offline_laplace_norm", **FSN_KW).load_synthetic(15).engine.stream_begin(1)


def test_dpcrn_real_checkpoint_forward_and_decode():
    """Real weights: vb_dpcrn_noncprs (1.0/1.0) and vb_dpcrn_cprs (0.5/2.0) on a full 4 s clip."""
    torch = _torch()
    from se_amd.models import dpcrn
    G = load_golden('dpcrn')
    ck = dict(load_golden('ckpt_vb_dpcrn_noncprs'))
    m = dpcrn(max_batch=2, max_samples=64000)
    m.load_state_dict(ck)
    y = m(torch.from_numpy(G['x']).cuda()).cpu().numpy()
    e = rms(y - G['y_real'])
    print('dpcrn real ckpt forward rms err', e, rms(G['y_real']))
    assert e < 2e-5 * max(rms(G['y_real']), 1.0)
    wav4 = synth.synth_clip(0, 'speech', 64000)
    x = torch.from_numpy(np.stack([wav4, synth.synth_clip(9, 'white', 64000)])).cuda()
    out = m.enhance_batch(x).cpu().numpy()
    e = rms(out[0] - G['enh_real'])
    print('dpcrn real ckpt decode rms err', e, rms(G['enh_real']))
    assert e < 1e-4 and e < 5e-4 * rms(G['enh_real'])
    mc = dpcrn(max_batch=1, max_samples=64000, p_in=0.5, p_out=2.0)
    mc.load_state_dict(dict(load_golden('ckpt_vb_dpcrn_cprs')))
    out = mc.enhance_batch(x[:1]).cpu().numpy()
    e = rms(out[0] - G['enh_real_cprs'])
    print('dpcrn real cprs ckpt decode rms err', e, rms(G['enh_real_cprs']))
    assert e < 1e-4 and e < 5e-4 * rms(G['enh_real_cprs'])


def test_strict_load_errors():
    _torch()
    from se_amd.models import crn_net
    from se_amd.engine import EngineError
    sd = synth.synth_state_dict(crn_net.state_dict_schema(), 1)
    bad = dict(sd)
    bad.pop('lstm.weight_hh_l1')
    with pytest.raises(RuntimeError):
        crn_net().load_state_dict(bad)
    bad = dict(sd)
    bad['lstm.weight_hh_l1'] = bad['lstm.weight_hh_l1'][:, :100]
    with pytest.raises(EngineError):
        crn_net().load_state_dict(bad)


@pytest.mark.parametrize('name,B', [('crn', 72), ('lstm', 70), ('gcrn', 136), ('fullsubnet', 130)])
def test_large_batch_matches_small_batch(name, B):
    """The weight-stationary cooperative LSTM kernel (k_lstm_coop.hip) slices sequences over workgroups and loops over
    16-sequence tiles when S > 16 * slices (ragged last tile, cell state in global scratch): a big ragged batch must
    reproduce, clip for clip, what the fixture-pinned B = 2 configuration gives."""
    torch = _torch()
    from se_amd.models import MODEL_CLASSES
    L = 4000
    clips = np.stack([synth.synth_clip(300 + i, 'speech' if i % 3 else 'white', L) for i in range(6)])
    wav = torch.from_numpy(clips[np.arange(B) % 6].copy()).cuda()
    big = MODEL_CLASSES[name](max_batch=B, max_samples=L).load_synthetic(SEEDS[name])
    yb = big.enhance_batch(wav).cpu().numpy()
    small = MODEL_CLASSES[name](max_batch=2, max_samples=L).load_synthetic(SEEDS[name])
    for i in (0, 2, 4):
        ys = small.enhance_batch(torch.from_numpy(clips[i:i + 2].copy()).cuda()).cpu().numpy()
        for j in (0, 1):
            for k in range(i + j, B, 6):
                e = rms(yb[k] - ys[j])
                assert e < 1e-5 * max(rms(ys[j]), 1e-4), (name, k, e, rms(ys[j]))     # fp32 tile shapes differ with the batch


@pytest.mark.parametrize('name', ['lstm', 'crn', 'gcrn', 'fullsubnet'])
def test_recurrence_forms_agree(name):
    """The recurrent layers run in different forms by batch size (k_lstm_coop.hip): one clip - dot products and, for LSTM / CRN,
    the whole stack as one wavefront launch with a tagged exchange; 2 ... 4 clips - K-split tiles, tagged exchange; 5 ... 16 - K-split
    tiles, flags; more - the sequence-sliced kernel.  A clip must decode to the same waveform in all of them (rounding of the
    differently ordered sums and of the 1-ulp tag aside), and the fixtures pin the small batches to the reference."""
    import torch
    from se_amd.models import MODEL_CLASSES
    L = 16000
    x = np.stack([synth.synth_clip(930 + b, 'speech', L) for b in range(20)])
    outs = {}
    for B in (1, 3, 8, 20):
        m = MODEL_CLASSES[name](max_batch=B, max_samples=L).load_synthetic(7)
        outs[B] = m.enhance_batch(torch.from_numpy(x[:B]).cuda()).cpu().numpy()[0]
    ref = outs[20]
    for B in (1, 3, 8):
        e = float(np.sqrt(np.mean((outs[B] - ref) ** 2)))
        print(name, 'batch', B, 'vs batch 20: rms err', e, 'rms', float(np.sqrt(np.mean(ref ** 2))))
        assert e < 1e-6 + 2e-5 * float(np.sqrt(np.mean(ref ** 2))), (name, B, e)


def _variant(name, n, **kw):
    from se_amd import models, models_new
    mod = models_new if '_new' in name else models
    if name.startswith('g2net'):
        return mod.gaf_base(3, 64, 2, 4, 4, [1, 2, 5, 9], 256 + 161 * 2, 256, 256, (2, 3), (1, 3), 64, 'cat', n, is_aux=False,
                            encoder_type='U2Net', tcm_type='full-band', **kw)
    return mod.TaylorSENet(cin=2, k1=(1, 3), k2=(2, 3), c=64, kd1=5, cd1=64, d_feat=256, dilations=[1, 2, 5, 9], p=2, fft_num=320,
                           order_num=n, intra_connect='cat', inter_connect='cat', is_causal=True, is_conformer=False, is_u2=True,
                           is_param_share=False, is_encoder_share=False, **kw)


@pytest.mark.parametrize('name,n,seed', [('g2net_s2', 2, 20), ('g2net_s4', 4, 20), ('g2net_new_s2', 2, 20), ('taylorsenet_o1', 1, 19),
                                         ('taylorsenet_o4', 4, 19), ('taylorsenet_new_o1', 1, 19)])
def test_stage_num_and_order_num_match_reference_fixtures(name, n, seed):
    """gaf_base(stage_num = n) (G2Net_VB/gaf_net_320.py:27,55-58) and TaylorSENet(order_num = n) (TaylorSENet/TaylorSENet.py:27,
    66-70) with values the decode scripts do not use (SE_CFG_REPEATS): forward and the compressed decode against fixtures of
    the imported reference built with the same value; a state dict of another count is rejected by the strict load."""
    torch = _torch()
    G = load_golden(name)
    m = _variant(name, n, max_batch=2, max_samples=8000, p_in=0.5, p_out=2.0).load_synthetic(seed)
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
    other = _variant(name, 3, max_batch=1, max_samples=8000)
    with pytest.raises(RuntimeError):
        other.load_state_dict(synth.synth_state_dict(m.state_dict_schema(), seed))
