"""Generate golden fixtures by IMPORTING the reference here (build container only).

TEST INFRASTRUCTURE.  Run:  python -m oracle.gen_golden [names...]
Needs /root/reference (never present on the GPU box); writes small .npz files
under tests/golden/.  Fixtures hold inputs and the reference's outputs only -
synthetic weights are regenerated from (schema, seed) by se_amd.synth.

What is imported from the reference, verbatim, per fixture:
  stft_*      torch.stft / torch.istft with the reference's arguments
  lstm_*      LSTM/LSTM.py             lstm_net
  crn_*       CRN/CRN.py               crn_net
  dpcrn_*     DPCRN/DPCRN.py           dpcrn   (+ real checkpoint vb_dpcrn_noncprs_model.pth)
  dccrn_*     DCCRN/DCCRN_cprs.py      DCCRN   (on top of oracle/_complexnn_recall.py - see there)
The decode-loop bodies (`enhance`) are re-stated around the imported model with
torch.stft/istft (legacy real-output API shimmed) exactly as SURVEY 8(c) says.
"""
import importlib
from collections import OrderedDict
import json
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)
import se_amd  # noqa: E402
from se_amd import synth  # noqa: E402


# ---------------------------------------------------------------- import shims
def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


# oracle/adopt_complexnn.py sets this to a supplied upstream complexnn.py (huyanxin/DeepComplexCRN); None = the recall
COMPLEXNN_PATH = None


def load_complexnn():
    """The module DCCRN/DCCRN_cprs.py:6 imports as `complexnn`: the supplied upstream file when COMPLEXNN_PATH is set,
    else oracle/_complexnn_recall.py (a restatement - DCCRN parity is then unpinned at this boundary)."""
    if COMPLEXNN_PATH is None:
        from oracle import _complexnn_recall
        return _complexnn_recall
    import importlib.util
    spec = importlib.util.spec_from_file_location('complexnn', COMPLEXNN_PATH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def install_stubs():
    """Import-time stubs for packages absent here; none is touched by forward."""
    for n in ('librosa', 'soundfile', 'h5py'):
        if n not in sys.modules:
            _stub(n)
    _stub('librosa.filters')
    _stub('pystoi', stoi=None)
    _stub('pystoi.stoi', stoi=None)
    _stub('ptflops', get_model_complexity_info=None)
    _stub('ptflops.flops_counter', get_model_complexity_info=None)
    _stub('torch_complex', ComplexTensor=None)
    _stub('show', show_model=lambda *a, **k: None, show_params=lambda *a, **k: None)
    sys.modules['complexnn'] = load_complexnn()
    _stub('conv_stft', ConvSTFT=None, ConviSTFT=None)


def import_ref(model_dir, module):
    """Import /root/reference/<model_dir>/<module>.py with that dir first on sys.path.
    config.py files mkdir ./BEST_MODEL etc. in the CWD -> run from a scratch dir."""
    install_stubs()
    scratch = '/tmp/se_golden_scratch'
    os.makedirs(scratch, exist_ok=True)
    cwd = os.getcwd()
    os.chdir(scratch)
    for m in ('Backup', 'config', 'data', 'Step2_config', module):
        sys.modules.pop(m, None)
    sys.path.insert(0, os.path.join(REF, model_dir))
    try:
        return importlib.import_module(module)
    finally:
        sys.path.pop(0)
        os.chdir(cwd)


def load_synth(model, seed):
    schema = synth.schema_of(model.state_dict())
    sd = synth.synth_state_dict(schema, seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.eval()
    return schema, sd


def t_stft(x, n_fft, hop, win):
    """Legacy real-output torch.stft ([B,F,T,2]) as the reference calls it."""
    return torch.view_as_real(torch.stft(x, n_fft, hop, win, window=torch.hann_window(win), return_complex=True))


FULL_ONLY = False      # `python -m oracle.gen_golden --full ...`: write only the full_<name>.npz (4 s clip) fixtures
FULL_SAMPLES = 64000   # BASELINE's clip: 16 kHz x 4 s
LONG_ONLY = False      # `python -m oracle.gen_golden --long ...`: write only the long<secs>_<name>.npz fixtures
# Clips longer than BASELINE's 4 s (VoiceBank+DEMAND reaches ~10-15 s): 10 s = 160 000 samples (T = 1001 / 1251 / 626) and
# just under 15 s with a length that is no multiple of any hop (239 987 samples: T = 1500 / 1876 / 938) - paths only long
# clips reach: the fused TCM kernel's T > 416 / T > 512 switches, cLN scans and FIR history past 401 frames, attention
# over > 1 160 keys, iSTFT windows.  (secs tag, samples, seed offset)
LONG_CLIPS = (('10', 160000, 1000), ('15', 239987, 2000))


def save_full(name, seed, enh_fn, kind='speech'):
    """One reference decode of a full-size (4 s) clip per network: tests/golden/full_<name>.npz holds the clip's synth
    seed and the reference's float32 output (256 kB); the clip itself is regenerated by se_amd.synth."""
    os.makedirs(GOLD, exist_ok=True)
    if LONG_ONLY:
        for tag, n, off in LONG_CLIPS:
            wav = synth.synth_clip(seed + off, kind, n)
            enh = np.asarray(enh_fn(wav), dtype=np.float32)
            path = os.path.join(GOLD, f'long{tag}_{name}.npz')
            np.savez_compressed(path, seed=np.int64(seed + off), n=np.int64(n), enh_cprs=enh)
            print(f'wrote {path}  {os.path.getsize(path) / 1024:.0f} kB')
        return
    wav = synth.synth_clip(seed, kind, FULL_SAMPLES)
    enh = np.asarray(enh_fn(wav), dtype=np.float32)
    path = os.path.join(GOLD, f'full_{name}.npz')
    np.savez_compressed(path, seed=np.int64(seed), n=np.int64(FULL_SAMPLES), enh4_cprs=enh)
    print(f'wrote {path}  {os.path.getsize(path) / 1024:.0f} kB')


def save(name, **arrs):
    if FULL_ONLY:
        return
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, name + '.npz')
    np.savez_compressed(path, **arrs)
    print(f'wrote {path}  {os.path.getsize(path) / 1024:.0f} kB')


def save_schema(name, schema):
    if FULL_ONLY:
        return
    with open(os.path.join(GOLD, f'schema_{name}.json'), 'w') as f:
        json.dump([[k, list(s), d] for k, (s, d) in schema.items()], f)


# ---------------------------------------------------------------- generators
def gen_stft():
    """torch.stft / torch.istft at the four reference geometries, float32 and float64."""
    out = {}
    for (n_fft, hop, win) in ((320, 160, 320), (512, 128, 512), (512, 256, 512), (512, 160, 400)):
        x = torch.from_numpy(synth.synth_batch(2, "speech", 4000, seed0=7))
        spec = torch.stft(x.double(), n_fft, hop, win, window=torch.hann_window(win, dtype=torch.float64), return_complex=True)
        y_len = torch.istft(spec, n_fft, hop, win, window=torch.hann_window(win, dtype=torch.float64), length=4000)
        y_nolen = torch.istft(spec, n_fft, hop, win, window=torch.hann_window(win, dtype=torch.float64))
        spec32 = torch.stft(x, n_fft, hop, win, window=torch.hann_window(win), return_complex=True)
        tag = f'{n_fft}_{hop}_{win}'
        out[f'x_{tag}'] = x.numpy()
        out[f'spec_{tag}'] = spec.numpy()
        out[f'spec32_{tag}'] = spec32.numpy()
        out[f'ylen_{tag}'] = y_len.numpy()
        out[f'ynolen_{tag}'] = y_nolen.numpy()
    save('stft', **out)


def _enhance_librosa_family(model, wav, in_kind, p_in=1.0, p_out=1.0):
    """Loop body shared by LSTM/CRN (in_kind='mag') and DPCRN ('ri'), with the librosa
    calls replaced by the verified-equivalent float64 torch.stft/istft."""
    x = torch.from_numpy(np.asarray(wav, dtype=np.float64))
    c = torch.sqrt(len(x) / torch.sum(x ** 2.0))
    x = x * c
    w = torch.hann_window(320, dtype=torch.float64)
    spec = torch.stft(x, 320, 160, 320, window=w, return_complex=True).T          # [T,F]
    mag, ph = spec.abs() ** p_in, spec.angle()
    with torch.no_grad():
        if in_kind == 'mag':
            est = model(mag.float()[None])[0].double() ** p_out
            de = est * torch.exp(1j * ph)
        else:
            mag32, ph32 = mag.float(), ph.float()
            feat = torch.stack((mag32 * torch.cos(ph32), mag32 * torch.sin(ph32)), dim=0)
            e = model(feat[None])
            emag = torch.norm(e, dim=1) ** p_out
            eph = torch.atan2(e[:, 1], e[:, 0])
            de = emag[0].double() * torch.exp(1j * eph[0].double())
    y = torch.istft(de.T, 320, 160, 320, window=w, length=len(x))
    return (y / c).numpy()


def gen_lstm():
    mod = import_ref('LSTM', 'LSTM')
    model = mod.lstm_net()
    schema, _ = load_synth(model, 11)
    save_schema('lstm', schema)
    rng = np.random.default_rng(5)
    x = np.abs(rng.standard_normal((2, 12, 161))).astype(np.float32)
    with torch.no_grad():
        y = model(torch.from_numpy(x)).numpy()
    wav = synth.synth_clip(3, 'speech', 4000)
    save('lstm', x=x, y=y, wav=wav, enh=_enhance_librosa_family(model, wav, 'mag'))
    save_full('lstm', 203, lambda w: _enhance_librosa_family(model, w, 'mag', 0.5, 2.0))


def gen_crn():
    mod = import_ref('CRN', 'CRN')
    model = mod.crn_net()
    schema, _ = load_synth(model, 12)
    save_schema('crn', schema)
    rng = np.random.default_rng(6)
    x = np.abs(rng.standard_normal((2, 10, 161))).astype(np.float32)
    with torch.no_grad():
        y = model(torch.from_numpy(x)).numpy()
    wav = synth.synth_clip(4, 'speech', 4000)
    save('crn', x=x, y=y, wav=wav, enh=_enhance_librosa_family(model, wav, 'mag'))
    save_full('crn', 204, lambda w: _enhance_librosa_family(model, w, 'mag', 0.5, 2.0))


def gen_dpcrn():
    mod = import_ref('DPCRN', 'DPCRN')
    model = mod.dpcrn()
    schema, _ = load_synth(model, 13)
    save_schema('dpcrn', schema)
    rng = np.random.default_rng(7)
    x = rng.standard_normal((2, 2, 9, 161)).astype(np.float32)
    with torch.no_grad():
        y = model(torch.from_numpy(x)).numpy()
    wav = synth.synth_clip(5, 'speech', 4000)
    enh = _enhance_librosa_family(model, wav, 'ri')
    # real checkpoint: the only real-weights anchor of the whole zoo
    ck = torch.load(os.path.join(REF, 'DPCRN/BEST_MODEL/vb_dpcrn_noncprs_model.pth'), map_location='cpu')
    model.load_state_dict(ck, strict=True)
    model.eval()
    with torch.no_grad():
        y_real = model(torch.from_numpy(x)).numpy()
    wav4 = synth.synth_clip(0, 'speech', 64000)
    enh_real = _enhance_librosa_family(model, wav4, 'ri')
    ckc = torch.load(os.path.join(REF, 'DPCRN/BEST_MODEL/vb_dpcrn_cprs_model.pth'), map_location='cpu')
    model.load_state_dict(ckc, strict=True)
    enh_real_cprs = _enhance_librosa_family(model, wav4, 'ri', 0.5, 2.0)
    save('dpcrn', x=x, y=y, wav=wav, enh=enh, y_real=y_real, enh_real=enh_real.astype(np.float32),
         enh_real_cprs=enh_real_cprs.astype(np.float32))
    if LONG_ONLY:           # long clips through the REAL compressed-spectrum checkpoint (loaded above)
        save_full('dpcrn', 0, lambda w: _enhance_librosa_family(model, w, 'ri', 0.5, 2.0))
    if FULL_ONLY:
        return
    # the checkpoints themselves, as data fixtures (fp32, int64 counters dropped -> re-added as zeros)
    for tag, c in (('noncprs', ck), ('cprs', ckc)):
        np.savez_compressed(os.path.join(GOLD, f'ckpt_vb_dpcrn_{tag}.npz'),
                            **{k: v.numpy() for k, v in c.items()})


def _enhance_dccrn(model, wav, p_in, p_out):
    """DCCRN/dccrn_decode_vb.py:25-62 around the imported model; librosa.istft -> float64 torch.istft."""
    feat_wav = np.asarray(wav, dtype=np.float64)
    c = np.sqrt(len(feat_wav) / np.sum(feat_wav ** 2.0))
    feat_wav = feat_wav * c
    wav_len = len(feat_wav)
    frame_num = int(np.ceil((wav_len - 512 + 512) / 128 + 1))
    fake = (frame_num - 1) * 128 + 512 - 512
    x = torch.FloatTensor(np.concatenate((feat_wav, np.zeros([fake - wav_len])), axis=0))
    feat_x = t_stft(x.unsqueeze(0), 512, 128, 512).permute(0, 3, 1, 2)
    mag, ph = torch.norm(feat_x, dim=1) ** p_in, torch.atan2(feat_x[:, 1], feat_x[:, 0])
    feat_x = torch.stack((mag * torch.cos(ph), mag * torch.sin(ph)), dim=1)
    with torch.no_grad():
        esti = model(feat_x)
    emag = torch.norm(esti, dim=1) ** p_out
    eph = torch.atan2(esti[:, 1], esti[:, 0])
    de = emag[0].double() * torch.exp(1j * eph[0].double())
    y = torch.istft(de, 512, 128, 512, window=torch.hann_window(512, dtype=torch.float64), length=len(x))
    return (y / c).numpy(), feat_x.numpy(), esti.numpy()


def gen_dccrn():
    mod = import_ref('DCCRN', 'DCCRN_cprs')
    model = mod.DCCRN(rnn_units=256, masking_mode='E', use_clstm=True, kernel_num=[32, 64, 128, 256, 256, 256])
    schema, _ = load_synth(model, 14)
    save_schema('dccrn', schema)
    rng = np.random.default_rng(8)
    x = rng.standard_normal((2, 2, 257, 7)).astype(np.float32)
    with torch.no_grad():
        y = model(torch.from_numpy(x)).numpy()
    wav = synth.synth_clip(6, 'speech', 4000)
    enh, _, _ = _enhance_dccrn(model, wav, 1.0, 1.0)
    enh_c, _, _ = _enhance_dccrn(model, wav, 0.5, 2.0)
    if LONG_ONLY:
        save_full('dccrn', 1, lambda w: _enhance_dccrn(model, w, 0.5, 2.0)[0])
        return
    wav4 = synth.synth_clip(1, 'speech', 64000)
    enh4, _, _ = _enhance_dccrn(model, wav4, 0.5, 2.0)
    save('dccrn', x=x, y=y, wav=wav, enh=enh, enh_cprs=enh_c, enh4_cprs=enh4.astype(np.float32))


def gen_dccrn_mask():
    """DCCRN(masking_mode='C' | 'R') - DCCRN/DCCRN_cprs.py:220-223: the two masks the class offers besides the decode script's
    'E' (same parameters, same key schema): forward + compressed-spectrum decode each."""
    if FULL_ONLY or LONG_ONLY:
        return
    mod = import_ref('DCCRN', 'DCCRN_cprs')
    rng = np.random.default_rng(18)
    x = rng.standard_normal((2, 2, 257, 7)).astype(np.float32)
    wav = synth.synth_clip(16, 'speech', 4000)
    out = {'x': x, 'wav': wav}
    for mode in ('C', 'R'):
        model = mod.DCCRN(rnn_units=256, masking_mode=mode, use_clstm=True, kernel_num=[32, 64, 128, 256, 256, 256])
        load_synth(model, 14)
        with torch.no_grad():
            out['y_' + mode] = model(torch.from_numpy(x)).numpy()
        out['enh_cprs_' + mode] = _enhance_dccrn(model, wav, 0.5, 2.0)[0]
    save('dccrn_mask', **out)


def _fsn_model(sequence_model="LSTM", norm_type="offline_laplace_norm"):
    install_stubs()
    scratch = '/tmp/se_golden_scratch'
    os.makedirs(scratch, exist_ok=True)
    for m in list(sys.modules):
        if m.startswith('fullsubnet_net_sa'):
            sys.modules.pop(m)
    sys.path.insert(0, os.path.join(REF, 'FullSubNet'))
    try:
        mod = importlib.import_module('fullsubnet_net_sa.model')
    finally:
        sys.path.pop(0)
    return mod.Model(sb_num_neighbors=15, fb_num_neighbors=0, num_freqs=257, look_ahead=2, sequence_model=sequence_model,
                     fb_output_activate_function="ReLU", sb_output_activate_function=None, fb_model_hidden_size=512,
                     sb_model_hidden_size=384, weight_init=True, norm_type=norm_type,
                     num_groups_in_drop_band=2)


def _enhance_fullsubnet(model, wav, p_in, p_out):
    """FullSubNet/fullsubnet_sa_decode_vb.py:37-72 around the imported model (B = 1)."""
    feat_wav = np.asarray(wav, dtype=np.float64)
    c = np.sqrt(len(feat_wav) / np.sum(feat_wav ** 2.0))
    feat_wav = feat_wav * c
    wav_len = len(feat_wav)
    x = torch.FloatTensor(feat_wav)
    feat_x = t_stft(x.unsqueeze(0), 512, 256, 512).permute(0, 3, 1, 2)
    mag = torch.norm(feat_x, dim=1) ** p_in
    ph = torch.atan2(feat_x[:, 1], feat_x[:, 0])
    feat_x = torch.stack((mag * torch.cos(ph), mag * torch.sin(ph)), dim=1)
    feat_mag = torch.norm(feat_x, dim=1, keepdim=True)
    with torch.no_grad():
        mask = model(feat_mag)
    mr, mi = mask[:, 0], mask[:, -1]
    fr, fi = feat_x[:, 0], feat_x[:, -1]
    er, ei = mr * fr - mi * fi, mr * fi + mi * fr
    esti = torch.stack((er, ei), dim=1)
    emag = torch.norm(esti, dim=1) ** p_out
    eph = torch.atan2(esti[:, 1], esti[:, 0])
    de = emag[0].double() * torch.exp(1j * eph[0].double())
    y = torch.istft(de, 512, 256, 512, window=torch.hann_window(512, dtype=torch.float64), length=wav_len)
    return (y / c).numpy()


def gen_fullsubnet():
    model = _fsn_model()
    schema, _ = load_synth(model, 15)
    save_schema('fullsubnet', schema)
    rng = np.random.default_rng(9)
    x = np.abs(rng.standard_normal((1, 1, 257, 9))).astype(np.float32)
    x2 = np.abs(rng.standard_normal((1, 1, 257, 9))).astype(np.float32)
    with torch.no_grad():
        y = model(torch.from_numpy(x)).numpy()
        y2 = model(torch.from_numpy(x2)).numpy()
    wav = synth.synth_clip(7, 'speech', 4000)
    save('fullsubnet', x=np.concatenate([x, x2]), y=np.concatenate([y, y2]), wav=wav,
         enh=_enhance_fullsubnet(model, wav, 1.0, 1.0), enh_cprs=_enhance_fullsubnet(model, wav, 0.5, 2.0))
    save_full('fullsubnet', 207, lambda w: _enhance_fullsubnet(model, w, 0.5, 2.0))


def gen_fullsubnet_cum():
    """Model(norm_type="cumulative_laplace_norm") - FullSubNet/fullsubnet_net_sa/base_model.py:212-240, :296-303: the causal
    normalisation (same parameters, same key schema as the decode script's model); forward + decode, and a 2 s clip for the
    frame-online mode."""
    if FULL_ONLY:
        return
    model = _fsn_model(norm_type="cumulative_laplace_norm")
    load_synth(model, 15)
    rng = np.random.default_rng(29)
    x = np.abs(rng.standard_normal((1, 1, 257, 11))).astype(np.float32)
    x2 = np.abs(rng.standard_normal((1, 1, 257, 11))).astype(np.float32)
    with torch.no_grad():
        y = model(torch.from_numpy(x)).numpy()
        y2 = model(torch.from_numpy(x2)).numpy()
    wav = synth.synth_clip(9, 'speech', 6000)
    wav2 = synth.synth_clip(10, 'speech', 32000)
    save('fullsubnet_cum', x=np.concatenate([x, x2]), y=np.concatenate([y, y2]), wav=wav,
         enh_cprs=_enhance_fullsubnet(model, wav, 0.5, 2.0), wav2=wav2, enh2_cprs=_enhance_fullsubnet(model, wav2, 0.5, 2.0))


def gen_fullsubnet_gru():
    """Model(sequence_model="GRU") - FullSubNet/fullsubnet_net_sa/sequence_model.py:36-43, the north star's GRU time step."""
    if FULL_ONLY:
        return
    model = _fsn_model("GRU")
    schema, _ = load_synth(model, 25)
    save_schema('fullsubnet_gru', schema)
    rng = np.random.default_rng(19)
    x = np.abs(rng.standard_normal((1, 1, 257, 9))).astype(np.float32)
    x2 = np.abs(rng.standard_normal((1, 1, 257, 9))).astype(np.float32)
    with torch.no_grad():
        y = model(torch.from_numpy(x)).numpy()
        y2 = model(torch.from_numpy(x2)).numpy()
    wav = synth.synth_clip(8, 'speech', 6000)
    save('fullsubnet_gru', x=np.concatenate([x, x2]), y=np.concatenate([y, y2]), wav=wav,
         enh_cprs=_enhance_fullsubnet(model, wav, 0.5, 2.0))


def gen_gcrn():
    mod = import_ref('GCRN', 'GCRN_noncprs')
    model = mod.Net()
    schema, _ = load_synth(model, 16)
    save_schema('gcrn', schema)
    rng = np.random.default_rng(10)
    x = rng.standard_normal((2, 2, 8, 161)).astype(np.float32)
    with torch.no_grad():
        y = model(torch.from_numpy(x)).numpy()
    wav = synth.synth_clip(8, 'speech', 4000)
    save('gcrn', x=x, y=y, wav=wav, enh=_enhance_librosa_family(model, wav, 'ri', 0.5, 2.0))
    save_full('gcrn', 208, lambda w: _enhance_librosa_family(model, w, 'ri', 0.5, 2.0))


def gen_ctsnet(ref_dir='CTSNet', tag='', X=6, R=3):
    m1 = import_ref(ref_dir, 'Step1_network').Step1_net()
    sch1, _ = load_synth(m1, 17)
    m2 = import_ref(ref_dir, 'Step2_network').Step2_net(X=X, R=R)
    sch2, _ = load_synth(m2, 18)
    if (X, R) == (6, 3):
        save_schema('cts_step1' + tag, sch1)
    save_schema('cts_step2' + tag, sch2)
    rng = np.random.default_rng(11)
    x1 = np.abs(rng.standard_normal((2, 40, 161))).astype(np.float32)
    x2 = rng.standard_normal((2, 4, 40, 161)).astype(np.float32)
    with torch.no_grad():
        y1 = m1(torch.from_numpy(x1)).numpy()
        y2 = m2(torch.from_numpy(x2)).numpy()

    def enh(wav, p_in, p_out):
        feat_wav = np.asarray(wav, dtype=np.float64)
        c = np.sqrt(len(feat_wav) / np.sum(feat_wav ** 2.0))
        feat_wav = feat_wav * c
        wav_len = len(feat_wav)
        frame_num = int(np.ceil((wav_len - 320 + 320) / 160 + 1))
        fake = (frame_num - 1) * 160
        xw = torch.FloatTensor(np.concatenate((feat_wav, np.zeros([fake - wav_len])), axis=0))
        feat_x_ = t_stft(xw.unsqueeze(0), 320, 160, 320).permute(0, 3, 2, 1)
        mag, ph = torch.norm(feat_x_, dim=1) ** p_in, torch.atan2(feat_x_[:, 1], feat_x_[:, 0])
        feat_x = torch.stack((mag * torch.cos(ph), mag * torch.sin(ph)), dim=1)
        with torch.no_grad():
            e1 = m1(torch.norm(feat_x, dim=1))
            s1 = torch.stack((e1 * torch.cos(ph), e1 * torch.sin(ph)), dim=1)
            s2 = m2(torch.cat((feat_x, s1), dim=1)) + s1
        emag = torch.norm(s2, dim=1) ** p_out
        eph = torch.atan2(s2[:, 1], s2[:, 0])
        de = emag[0].double() * torch.exp(1j * eph[0].double())          # [T,F]
        y = torch.istft(de.T, 320, 160, 320, window=torch.hann_window(320, dtype=torch.float64))[:wav_len]
        return (y / c).numpy()
    wav = synth.synth_clip(9, 'speech', 8000)
    if (X, R) != (6, 3):      # (another constructor value: the second stage's fixture and the chained decode only)
        save('ctsnet' + tag, x2=x2, y2=y2, wav=wav, enh_cprs=enh(wav, 0.5, 2.0))
        return
    save('ctsnet' + tag, x1=x1, y1=y1, x2=x2, y2=y2, wav=wav, enh=enh(wav, 1.0, 1.0), enh_cprs=enh(wav, 0.5, 2.0))
    save_full('ctsnet' + tag, 209, lambda w: enh(w, 0.5, 2.0))


def gen_taylorsenet(ref_dir='TaylorSENet', tag='', order_num=3):
    mod = import_ref(ref_dir, 'TaylorSENet')
    model = mod.TaylorSENet(cin=2, k1=(1, 3), k2=(2, 3), c=64, kd1=5, cd1=64, d_feat=256, dilations=[1, 2, 5, 9], p=2,
                            fft_num=320, order_num=order_num, intra_connect='cat', inter_connect='cat', is_causal=True,
                            is_conformer=False, is_u2=True, is_param_share=False, is_encoder_share=False)
    schema, _ = load_synth(model, 19)
    save_schema('taylorsenet' + tag, schema)
    rng = np.random.default_rng(12)
    x = rng.standard_normal((2, 2, 30, 161)).astype(np.float32)
    with torch.no_grad():
        y = model(torch.from_numpy(x)).numpy()

    def enh(wav, p_in, p_out):
        feat_wav = np.asarray(wav, dtype=np.float64)
        c = np.sqrt(len(feat_wav) / np.sum(feat_wav ** 2.0))
        feat_wav = feat_wav * c
        wav_len = len(feat_wav)
        frame_num = int(np.ceil((wav_len - 320 + 320) / 160 + 1))
        xw = torch.FloatTensor(np.concatenate((feat_wav, np.zeros([(frame_num - 1) * 160 - wav_len])), axis=0))
        feat_x = t_stft(xw.unsqueeze(0), 320, 160, 320).permute(0, 3, 2, 1)
        mag, ph = torch.norm(feat_x, dim=1) ** p_in, torch.atan2(feat_x[:, -1], feat_x[:, 0])
        fc = torch.stack((mag * torch.cos(ph), mag * torch.sin(ph)), dim=1)
        with torch.no_grad():
            e = model(fc)
        emag, eph = torch.norm(e, dim=1) ** p_out, torch.atan2(e[:, -1], e[:, 0])
        de = emag[0].double() * torch.exp(1j * eph[0].double())
        y = torch.istft(de.T, 320, 160, 320, window=torch.hann_window(320, dtype=torch.float64), length=wav_len)
        return (y / c).numpy()
    wav = synth.synth_clip(10, 'speech', 6000)
    save('taylorsenet' + tag, x=x, y=y, wav=wav, enh=enh(wav, 1.0, 1.0), enh_cprs=enh(wav, 0.5, 2.0))
    if order_num == 3:      # (the other constructor values: the small fixture only)
        save_full('taylorsenet' + tag, 210, lambda w: enh(w, 0.5, 2.0))


def gen_g2net(ref_dir='G2Net_VB', tag='', stage_num=3):
    install_stubs()
    mod = import_ref(ref_dir, 'gaf_net_320')
    model = mod.gaf_base(3, 64, 2, 4, 4, [1, 2, 5, 9], 256 + 161 * 2, 256, 256, (2, 3), (1, 3), 64, 'cat', stage_num,
                         is_aux=False, encoder_type='U2Net', tcm_type='full-band')
    schema, _ = load_synth(model, 20)
    save_schema('g2net' + tag, schema)
    rng = np.random.default_rng(13)
    x = rng.standard_normal((2, 2, 30, 161)).astype(np.float32)
    with torch.no_grad():
        ys = model(torch.from_numpy(x))
    y = ys[-1].numpy()
    y0 = ys[0].numpy()

    def enh(wav, p_in, p_out):
        xw = np.asarray(wav, dtype=np.float64)
        c = np.sqrt(np.sum(xw ** 2.0) / len(xw))
        xt = torch.from_numpy(xw / c)
        w = torch.hann_window(320, dtype=torch.float64)
        spec = torch.stft(xt, 320, 160, 320, window=w, return_complex=True).T
        mag, ph = spec.abs() ** p_in, spec.angle()
        feat = torch.stack(((mag * torch.cos(ph)).float(), (mag * torch.sin(ph)).float()), dim=0)
        with torch.no_grad():
            e = model(feat[None])[-1][0].permute(0, 2, 1)
        emag, eph = torch.norm(e, dim=0) ** p_out, torch.atan2(e[1], e[0])
        de = (emag * torch.cos(eph)).double() + 1j * (emag * torch.sin(eph)).double()
        yy = torch.istft(de.T, 320, 160, 320, window=w, length=len(xt))
        return (yy * c).numpy()
    wav = synth.synth_clip(11, 'speech', 6000)
    save('g2net' + tag, x=x, y=y, y0=y0, wav=wav, enh=enh(wav, 1.0, 1.0), enh_cprs=enh(wav, 0.5, 2.0))
    if stage_num == 3:      # (the other constructor values: the small fixture only)
        save_full('g2net' + tag, 211, lambda w: enh(w, 0.5, 2.0))


def gen_uformer():
    """Uformer.forward hard-codes .cuda() and the legacy real-output torch.stft/istft API: shimmed."""
    install_stubs()
    for m in ('uformer', 'dilated_dualpath_conformer', 'trans'):
        sys.modules.pop(m, None)
    mod = import_ref('Uformer', 'uformer')
    model = mod.Uformer()
    sd_full = model.state_dict()
    keep = OrderedDict((k, v) for k, v in sd_full.items() if not (k.startswith('stft.') or k.startswith('istft.')))
    schema = synth.schema_of(keep)
    sdn = synth.synth_state_dict(schema, 21)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sdn.items()}, strict=False)
    model.eval()
    save_schema('uformer', schema)
    o_stft, o_istft, o_cuda = torch.stft, torch.istft, torch.Tensor.cuda
    torch.stft = lambda x, **k: torch.view_as_real(o_stft(x, return_complex=True, **k))
    torch.istft = lambda x, **k: o_istft(torch.view_as_complex(x.contiguous()), **k)
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        wav = synth.synth_clip(12, 'speech', 4000)
        c = np.sqrt(len(wav) / np.sum(wav.astype(np.float64) ** 2.0))
        xw = torch.FloatTensor(wav.astype(np.float64) * c)
        with torch.no_grad():
            out, _, cplx, _ = model(xw[None], xw[None])
        enh = (out[0].numpy() / c)
        # the full 4-tuple with a source that differs from the input (uformer.py:182-194, :287): src = another clip
        src = torch.FloatTensor(synth.synth_clip(13, 'speech', 4000).astype(np.float64) * c)
        with torch.no_grad():
            out2, src_out, cplx2, src_cplx = model(xw[None], src[None])
        assert torch.equal(out2, out) and torch.equal(cplx2, cplx)
        save('uformer', wav=wav, enh=enh, cplx=cplx.numpy(), src_wav=src_out.numpy(), src_cplx=src_cplx.numpy())

        def enh_full(w):
            cc = np.sqrt(len(w) / np.sum(w.astype(np.float64) ** 2.0))
            xx = torch.FloatTensor(w.astype(np.float64) * cc)
            with torch.no_grad():
                o = model(xx[None], xx[None])[0]
            return o[0].numpy() / cc
        save_full('uformer', 212, enh_full)
    finally:
        torch.stft, torch.istft, torch.Tensor.cuda = o_stft, o_istft, o_cuda


def gen_ctsnet_new():
    """CTSNet_new: CTSNet with every InstanceNorm replaced by CumulativeLayerNorm (Step1_network.py:213-286)."""
    gen_ctsnet('CTSNet_new', '_new')


def gen_taylorsenet_new():
    gen_taylorsenet('TaylorSENet_new', '_new')


def gen_g2net_new():
    gen_g2net('G2Net_new', '_new')


def gen_repeat_counts():
    """Constructor values the decode scripts do not use: gaf_base(stage_num = 2 / 4) (gaf_net_320.py:27,55-58) and
    TaylorSENet(order_num = 1 / 4) (TaylorSENet.py:27,66-70); one cLN flavour of each."""
    gen_g2net(tag='_s2', stage_num=2)
    gen_g2net(tag='_s4', stage_num=4)
    gen_g2net('G2Net_new', '_new_s2', stage_num=2)
    gen_taylorsenet(tag='_o1', order_num=1)
    gen_taylorsenet(tag='_o4', order_num=4)
    gen_taylorsenet('TaylorSENet_new', '_new_o1', order_num=1)
    # Step2_net(X, R) (CTSNet/Step2_network.py:13-21)
    gen_ctsnet(tag='_x4r2', X=4, R=2)
    gen_ctsnet('CTSNet_new', '_new_x5r4', X=5, R=4)


GENS = {'repeat_counts': gen_repeat_counts, 'stft': gen_stft, 'dccrn_mask': gen_dccrn_mask, 'fullsubnet_cum': gen_fullsubnet_cum, 'fullsubnet_gru': gen_fullsubnet_gru, 'ctsnet_new': gen_ctsnet_new, 'taylorsenet_new': gen_taylorsenet_new, 'g2net_new': gen_g2net_new, 'uformer': gen_uformer, 'g2net': gen_g2net, 'taylorsenet': gen_taylorsenet, 'ctsnet': gen_ctsnet, 'gcrn': gen_gcrn, 'fullsubnet': gen_fullsubnet, 'lstm': gen_lstm, 'crn': gen_crn, 'dpcrn': gen_dpcrn, 'dccrn': gen_dccrn}

if __name__ == '__main__':
    torch.set_num_threads(8)
    names = sys.argv[1:]
    if names and names[0] == '--full':
        FULL_ONLY, names = True, names[1:]
    if names and names[0] == '--long':
        FULL_ONLY, LONG_ONLY, names = True, True, names[1:]
    names = names or list(GENS)
    for n in names:
        GENS[n]()
