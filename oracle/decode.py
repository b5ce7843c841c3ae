"""Numpy restatement of the per-utterance body of each `enhance(args)` loop.

TEST INFRASTRUCTURE (see oracle/__init__.py).  One function per decode script;
each takes the model state dict and ONE float waveform (what `sf.read`
returns) and gives back the float waveform the script hands to `sf.write`
(before PCM16 quantisation).  `p_in` / `p_out` are the magnitude exponents the
scripts hard-code (1.0/1.0 "noncprs", 0.5/2.0 "cprs").

The reference feeds the network float32 (`torch.FloatTensor`); `net_dtype`
selects that (np.float32) or a float64 "truth" run.
"""
import numpy as np
from . import stft as S
from . import models as M


def _frontend_librosa(wav, n_fft, hop):
    wav = np.asarray(wav, dtype=np.float64)
    c = S.rms_scale(wav)
    x = wav * c
    spec = S.stft(x, n_fft, hop).T                     # [T,F]  (librosa.stft(...).T)
    return c, x, spec


def enhance_lstm(sd, wav, p_in=1.0, p_out=1.0, net_dtype=np.float32, fwd=M.lstm_net_forward):
    """LSTM/lstm_decode_vb.py:33-52 (and CRN/crn_decode_vb.py:33-52 with fwd=crn)."""
    c, x, spec = _frontend_librosa(wav, 320, 160)
    mag, ph = np.abs(spec) ** p_in, np.angle(spec)     # :38
    est = fwd(sd, mag[None].astype(net_dtype))[0].astype(np.float64)   # :41-45
    est = est ** p_out                                 # :47
    de = est * np.exp(1j * ph)                         # :49
    y = S.istft(de.T, 320, 160, length=len(x))         # :50-51
    return y / c                                       # :52


def enhance_crn(sd, wav, p_in=1.0, p_out=1.0, net_dtype=np.float32):
    return enhance_lstm(sd, wav, p_in, p_out, net_dtype, fwd=M.crn_net_forward)


def enhance_dpcrn(sd, wav, p_in=1.0, p_out=1.0, net_dtype=np.float32):
    """DPCRN/dpcrn_decode_vb.py:33-60."""
    c, x, spec = _frontend_librosa(wav, 320, 160)
    mag, ph = np.abs(spec) ** p_in, np.angle(spec)                       # :41
    mag32, ph32 = mag.astype(net_dtype), ph.astype(net_dtype)            # torch.FloatTensor :43-44
    feat = np.stack([mag32 * np.cos(ph32), mag32 * np.sin(ph32)], 0)     # :45
    est = M.dpcrn_forward(sd, feat[None])                                # :47
    emag = np.sqrt(est[:, 0] ** 2 + est[:, 1] ** 2)                      # :48
    eph = np.arctan2(est[:, 1], est[:, 0])                               # :49
    emag = emag ** p_out                                                 # :53
    de = emag[0].astype(np.float64) * np.exp(1j * eph[0].astype(np.float64))   # :55-57
    y = S.istft(de.T, 320, 160, length=len(x))                           # :58-59
    return y / c


def enhance_dccrn(sd, wav, p_in=1.0, p_out=1.0, net_dtype=np.float32, masking_mode='E'):
    """DCCRN/dccrn_decode_vb.py:25-62.  Output length = padded length (:59-60)."""
    wav = np.asarray(wav, dtype=np.float64)
    c = S.rms_scale(wav)                                                 # :27
    x = S.pad_to_hop(wav * c, 512, 128).astype(net_dtype)                # :28-35 (FloatTensor)
    spec = S.stft(x, 512, 128)                                           # [F,T] :37-38
    re, im = spec.real.astype(net_dtype), spec.imag.astype(net_dtype)
    mag = np.sqrt(re ** 2 + im ** 2) ** p_in                             # :40
    ph = np.arctan2(im, re)
    feat = np.stack([mag * np.cos(ph), mag * np.sin(ph)], 0)[None]       # :42  [1,2,F,T]
    est = M.dccrn_forward(sd, feat, masking_mode=masking_mode)           # :44
    emag = np.sqrt(est[:, 0] ** 2 + est[:, 1] ** 2)                      # :45
    eph = np.arctan2(est[:, 1], est[:, 0])                               # :46
    emag = emag ** p_out                                                 # :48
    de = emag[0].astype(np.float64) * np.exp(1j * eph[0].astype(np.float64))   # :56-58
    y = S.istft(de, 512, 128, length=len(x))                             # :59-60
    return y / c                                                         # :62


ENHANCE = {
    'lstm': enhance_lstm,
    'crn': enhance_crn,
    'dpcrn': enhance_dpcrn,
    'dccrn': enhance_dccrn,
}


def enhance_fullsubnet(sd, wav, p_in=1.0, p_out=1.0, net_dtype=np.float32, norm_type='offline_laplace_norm'):
    """FullSubNet/fullsubnet_sa_decode_vb.py:37-72 (the computed tail pad :42-45 is never applied)."""
    wav = np.asarray(wav, dtype=np.float64)
    c = S.rms_scale(wav)                                                 # :38
    x = (wav * c).astype(net_dtype)                                      # :39,45 FloatTensor
    spec = S.stft(x, 512, 256)                                           # :46-47
    re, im = spec.real.astype(net_dtype), spec.imag.astype(net_dtype)
    mag = np.sqrt(re ** 2 + im ** 2) ** p_in                             # :50
    ph = np.arctan2(im, re)
    fr, fi = mag * np.cos(ph), mag * np.sin(ph)                          # :52
    fmag = np.sqrt(fr ** 2 + fi ** 2)[None, None]                        # :54
    mask = M.fullsubnet_forward(sd, fmag, norm_type=norm_type)           # :56
    mr, mi = mask[0, 0], mask[0, 1]
    er = mr * fr - mi * fi                                               # :59-60
    ei = mr * fi + mi * fr
    emag = np.sqrt(er ** 2 + ei ** 2) ** p_out                           # :64
    eph = np.arctan2(ei, er)
    de = emag.astype(np.float64) * np.exp(1j * eph.astype(np.float64))
    y = S.istft(de, 512, 256, length=len(x))                             # :69
    return y / c


ENHANCE['fullsubnet'] = enhance_fullsubnet


def enhance_gcrn(sd, wav, p_in=0.5, p_out=2.0, net_dtype=np.float32):
    """GCRN/gcrn_decode_vb.py:34-58 (checked in with the compressed exponents 0.5 / 2.0, :40,:51)."""
    c, x, spec = _frontend_librosa(wav, 320, 160)
    mag, ph = np.abs(spec) ** p_in, np.angle(spec)                       # :40
    mag32, ph32 = mag.astype(net_dtype), ph.astype(net_dtype)
    feat = np.stack([mag32 * np.cos(ph32), mag32 * np.sin(ph32)], 0)     # :44
    est = M.gcrn_forward(sd, feat[None])                                 # :46
    emag = np.sqrt(est[:, 0] ** 2 + est[:, 1] ** 2) ** p_out             # :47,:51
    eph = np.arctan2(est[:, 1], est[:, 0])                               # :48
    de = emag[0].astype(np.float64) * np.exp(1j * eph[0].astype(np.float64))
    y = S.istft(de.T, 320, 160, length=len(x))                           # :56-57
    return y / c


ENHANCE['gcrn'] = enhance_gcrn


def enhance_ctsnet(sd1, sd2, wav, p_in=1.0, p_out=1.0, net_dtype=np.float32):
    """CTSNet/two_stage_com_decode_vb.py:62-96: two chained models, torch.stft 320/160, istft without `length`."""
    wav = np.asarray(wav, dtype=np.float64)
    c = S.rms_scale(wav)                                                 # :63
    L = len(wav)
    x = S.pad_to_hop(wav * c, 320, 160).astype(net_dtype)                # :65-69
    spec = S.stft(x, 320, 160).T                                         # [T,F]   :70-71 permute(0,3,2,1)
    re, im = spec.real.astype(net_dtype), spec.imag.astype(net_dtype)
    mag = np.sqrt(re ** 2 + im ** 2) ** p_in                             # :73
    ph = np.arctan2(im, re)
    feat = np.stack([mag * np.cos(ph), mag * np.sin(ph)], 0)[None]       # :75  [1,2,T,F]
    est1 = M.cts_step1_forward(sd1, np.sqrt(feat[:, 0] ** 2 + feat[:, 1] ** 2))   # :79
    s1 = np.stack([est1 * np.cos(ph), est1 * np.sin(ph)], 1)             # :80-81
    s2 = M.cts_step2_forward(sd2, np.concatenate([feat, s1], 1)) + s1    # :82-84
    emag = np.sqrt(s2[:, 0] ** 2 + s2[:, 1] ** 2) ** p_out               # :87
    eph = np.arctan2(s2[:, 1], s2[:, 0])
    de = emag[0].astype(np.float64) * np.exp(1j * eph[0].astype(np.float64))
    y = S.istft(de.T, 320, 160)[:L]                                      # :93-95
    return y / c


def enhance_taylorsenet(sd, wav, p_in=1.0, p_out=1.0, net_dtype=np.float32):
    """TaylorSENet/taylorsenet_decode_vb.py:26-51 (istft with length=wav_len)."""
    wav = np.asarray(wav, dtype=np.float64)
    c = S.rms_scale(wav)
    L = len(wav)
    x = S.pad_to_hop(wav * c, 320, 160).astype(net_dtype)                # :30-35
    spec = S.stft(x, 320, 160).T                                         # [T,F]
    re, im = spec.real.astype(net_dtype), spec.imag.astype(net_dtype)
    mag = np.sqrt(re ** 2 + im ** 2) ** p_in                             # :40
    ph = np.arctan2(im, re)
    feat = np.stack([mag * np.cos(ph), mag * np.sin(ph)], 0)[None]
    est = M.taylorsenet_forward(sd, feat)                                # :42
    emag = np.sqrt(est[:, 0] ** 2 + est[:, 1] ** 2) ** p_out             # :44
    eph = np.arctan2(est[:, -1], est[:, 0])
    de = emag[0].astype(np.float64) * np.exp(1j * eph[0].astype(np.float64))
    y = S.istft(de.T, 320, 160, length=L)                                # :48
    return y / c


ENHANCE['taylorsenet'] = enhance_taylorsenet


def enhance_g2net(sd, wav, p_in=1.0, p_out=1.0, net_dtype=np.float32):
    """G2Net_VB/com_decode.py:39-88: normalises with x / c, c = RMS (:43-44) and restores with * c (:88)."""
    wav = np.asarray(wav, dtype=np.float64)
    c = np.sqrt(np.sum(wav ** 2.0) / len(wav))
    x = wav / c
    spec = S.stft(x, 320, 160).T                                          # :49
    mag, ph = np.abs(spec) ** p_in, np.angle(spec)                        # :53
    feat = np.stack([(mag * np.cos(ph)).astype(net_dtype), (mag * np.sin(ph)).astype(net_dtype)], 0)   # :61
    est = M.g2net_forward(sd, feat[None])[-1][0]                          # :66-69  [2,161,T]
    est = np.transpose(est, (0, 2, 1))                                    # :74
    emag = np.sqrt(est[0] ** 2 + est[1] ** 2) ** p_out                    # :76
    eph = np.arctan2(est[1], est[0])
    de = emag.astype(np.float64) * np.exp(1j * eph.astype(np.float64))
    y = S.istft(de.T, 320, 160, length=len(x))                            # :86-87
    return y * c


ENHANCE['g2net'] = enhance_g2net


def enhance_uformer(sd, wav, net_dtype=np.float32):
    """Uformer/uformer_decode_vb.py:34-62 + the STFT/iSTFT inside Uformer.forward (uformer.py:178, 276):
    torch.stft(n_fft=512, hop=160, win=400), istft without `length` -> 160*floor(L/160) samples."""
    wav = np.asarray(wav, dtype=np.float64)
    c = S.rms_scale(wav)
    x = (wav * c).astype(net_dtype)                                      # :39 FloatTensor
    spec = S.stft(x, 512, 160, 400)
    re, im = spec.real.astype(net_dtype)[None], spec.imag.astype(net_dtype)[None]
    er, ei = M.uformer_core(sd, re, im)
    de = er[0].astype(np.float64) + 1j * ei[0].astype(np.float64)
    y = S.istft(de, 512, 160, 400)
    return y / c


def uformer_forward4(sd, inputs, src, net_dtype=np.float32):
    """`output, src, output_cplx, src_cplx = model(inputs, src)` for ONE pair of waveforms (Uformer/uformer.py:172-287):
    output = istft of the estimate (:276), src = istft(stft(src)) (:186), output_cplx [2,257,T] the RI estimate (:264-286),
    src_cplx [2,257,T] = |S| e^{j angle S} of the source's STFT with the reference's clamp / EPS (:187-194)."""
    x = np.asarray(inputs, dtype=net_dtype)
    spec = S.stft(x, 512, 160, 400)
    er, ei = M.uformer_core(sd, spec.real.astype(net_dtype)[None], spec.imag.astype(net_dtype)[None])
    out = S.istft(er[0].astype(np.float64) + 1j * ei[0].astype(np.float64), 512, 160, 400)
    ss = S.stft(np.asarray(src, dtype=net_dtype), 512, 160, 400)
    sr, si = ss.real.astype(net_dtype), ss.imag.astype(net_dtype)
    src_out = S.istft(sr.astype(np.float64) + 1j * si.astype(np.float64), 512, 160, 400)
    eps = np.float32(np.finfo(np.float32).eps)
    mag = np.sqrt(np.maximum(sr ** 2 + si ** 2, eps))
    ph = np.arctan2(si + eps, sr)
    return out, src_out, np.stack([er[0], ei[0]]), np.stack([mag * np.cos(ph), mag * np.sin(ph)])


ENHANCE['uformer'] = enhance_uformer
