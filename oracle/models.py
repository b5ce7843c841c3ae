"""Numpy restatement of the reference `nn.Module.forward`s on the hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Each function takes the model's
flat state dict `sd` (name -> numpy array, the key schema of the reference's
`state_dict()`) and the input in the reference's layout and
returns what the reference `forward` returns.  Compute dtype = input dtype.
"""
import numpy as np
from . import nnops as nn


def _bn(sd, p, x):
    return nn.batchnorm(x, sd[p + 'weight'], sd[p + 'bias'], sd[p + 'running_mean'], sd[p + 'running_var'])


# ----------------------------------------------------------------------------
# LSTM   (reference LSTM/LSTM.py:14-28)
# ----------------------------------------------------------------------------
def lstm_net_forward(sd, x):
    """x [B,T,161] magnitude -> [B,T,161] enhanced magnitude (mapping)."""
    # LSTM.py:25  BatchNorm1d over the feature dim (permute to [B,161,T] and back)
    x = _bn(sd, 'bn.', np.swapaxes(x, 1, 2))
    x = np.swapaxes(x, 1, 2)
    x = nn.lstm(x, sd, 'lstm1.', 1, batch_first=True)          # LSTM.py:26
    x = nn.lstm(x, sd, 'lstm2.', 2, batch_first=True)          # LSTM.py:27
    x = nn.softplus(nn.linear(x, sd['fc.0.weight'], sd['fc.0.bias']))   # LSTM.py:20-22,28
    return x


# ----------------------------------------------------------------------------
# CRN   (reference CRN/CRN.py:16-117)
# ----------------------------------------------------------------------------
def _causal_conv_tf(sd, p, x):
    """ConstantPad2d((0,0,1,0)) + Conv2d(k=(2,3), s=(1,2))   CRN.py:38-42, DPCRN.py:97-101."""
    x = np.pad(x, ((0, 0), (0, 0), (1, 0), (0, 0)))
    return nn.conv2d(x, sd[p + 'weight'], sd[p + 'bias'], stride=(1, 2))


def crn_net_forward(sd, x):
    """x [B,T,161] -> [B,T,161]."""
    x = x[:, None]                                              # CRN.py:24
    B, _, T, _ = x.shape
    skips = []
    for i in range(5):                                          # Encoder CRN.py:35-71
        x = _causal_conv_tf(sd, f'en.en_module.{i}.1.', x)
        x = nn.elu(_bn(sd, f'en.en_module.{i}.2.', x))
        skips.append(x)
    x = np.transpose(x, (0, 2, 1, 3)).reshape(B, T, -1)         # CRN.py:27-28
    x = nn.lstm(x, sd, 'lstm.', 2, batch_first=True)            # CRN.py:29
    x = np.transpose(x.reshape(B, T, 256, 4), (0, 2, 1, 3))     # CRN.py:30-31
    for i in range(5):                                          # Decoder CRN.py:73-109
        x = np.concatenate([x, skips[-(i + 1)]], axis=1)        # CRN.py:107
        p = f'de.de_module.{i}.'
        x = nn.conv_transpose2d(x, sd[p + '0.weight'], sd[p + '0.bias'], stride=(1, 2))
        if i == 3:                                              # de4: ConstantPad2d((1,0,0,0)) CRN.py:93-95
            x = np.pad(x, ((0, 0), (0, 0), (0, 0), (1, 0)))
            bn = p + '3.'
        else:
            bn = p + '2.'
        x = x[:, :, :-1, :]                                     # Chomp_T(1) CRN.py:112-117
        x = _bn(sd, bn, x)
        x = nn.softplus(x) if i == 4 else nn.elu(x)             # CRN.py:99-102
    return x[:, 0]                                              # squeeze CRN.py:109


# ----------------------------------------------------------------------------
# DPCRN   (reference DPCRN/DPCRN.py:16-174)
# ----------------------------------------------------------------------------
def _dprnn(sd, x):
    """DPRNN.forward DPCRN.py:59-92.  x [B,C=128,T,F=4]."""
    B, C, T, F = x.shape
    xp = np.transpose(x, (0, 2, 3, 1))                          # [B,T,F,C]
    out = xp.reshape(-1, F, C)                                  # [B*T,F,C]
    out = nn.lstm(out, sd, 'dprnn.intra_rnn.', 2, bidirectional=True, batch_first=True)
    out = nn.linear(out, sd['dprnn.intra_fc.weight'], sd['dprnn.intra_fc.bias'])
    out = out.reshape(B, -1, F, C)
    out = nn.layernorm(out, sd['dprnn.ln1.weight'], sd['dprnn.ln1.bias'], 2)
    intra = out + xp
    out = np.transpose(intra, (0, 2, 1, 3)).reshape(-1, T, C)   # [B*F,T,C]
    out = nn.lstm(out, sd, 'dprnn.inter_rnn.', 2, batch_first=True)
    out = nn.linear(out, sd['dprnn.inter_fc.weight'], sd['dprnn.inter_fc.bias'])
    out = np.transpose(out.reshape(B, -1, T, C), (0, 2, 1, 3))  # [B,T,F,C]
    out = nn.layernorm(out, sd['dprnn.ln2.weight'], sd['dprnn.ln2.bias'], 2)
    out = out + intra
    return np.transpose(out, (0, 3, 1, 2))


def dpcrn_forward(sd, inpt):
    """inpt [B,2,T,161] (RI) -> [B,2,T,161] masked RI.  DPCRN.py:23-42."""
    x = inpt
    skips = []
    for i in range(5):                                          # Encoder DPCRN.py:94-130
        x = _causal_conv_tf(sd, f'en.en_module.{i}.1.', x)
        x = nn.prelu(_bn(sd, f'en.en_module.{i}.2.', x), sd[f'en.en_module.{i}.3.weight'])
        skips.append(x)
    x = _dprnn(sd, x)                                           # DPCRN.py:28-29 (same weights twice)
    x = _dprnn(sd, x)
    for i in range(5):                                          # Decoder DPCRN.py:132-166
        x = np.concatenate([x, skips[-(i + 1)]], axis=1)
        p = f'de.de_module.{i}.'
        x = nn.conv_transpose2d(x, sd[p + '0.weight'], sd[p + '0.bias'], stride=(1, 2))
        off = 1
        if i == 3:                                              # de4 pad1 DPCRN.py:151-154
            x = np.pad(x, ((0, 0), (0, 0), (0, 0), (1, 0)))
            off = 2
        x = x[:, :, :-1, :]
        if i < 4:
            x = nn.prelu(_bn(sd, f'{p}{off + 1}.', x), sd[f'{p}{off + 2}.weight'])
    mr, mi = x[:, 0], x[:, 1]
    xr, xi = inpt[:, 0], inpt[:, 1]
    return np.stack([xr * mr - xi * mi, xr * mi + xi * mr], axis=1)   # DPCRN.py:33-42


# ----------------------------------------------------------------------------
# DCCRN   (reference DCCRN/DCCRN_cprs.py:8-226; operators of the ABSENT
# third-party `complexnn.py` restated from upstream huyanxin/DeepComplexCRN -
# PARITY UNPINNED at that boundary, see oracle/_complexnn_recall.py)
# ----------------------------------------------------------------------------
# The two conventions DCCRN_cprs.py does not determine, mirrored from include/se_engine.h
# (SE_CFG_DCCRN_BIAS_PER_PART, SE_CFG_DCCRN_PLAIN_CAT): default = the published upstream as restated in _complexnn_recall.py
DCCRN_BIAS_PER_PART, DCCRN_PLAIN_CAT = 2, 4


def _cplx_combine(f, r, i, wr, br, wi, bi, variant):
    if variant & DCCRN_BIAS_PER_PART:
        z = np.zeros_like(br)
        return f(r, wr, br) - f(i, wi, z), f(r, wi, bi) + f(i, wr, z)
    return f(r, wr, br) - f(i, wi, bi), f(r, wi, bi) + f(i, wr, br)


def _cplx_conv2d(sd, p, x, pad_f, pad_t, variant=0):
    """complexnn.ComplexConv2d(causal=True, complex_axis=1) as called at
    DCCRN_cprs.py:66-72: time padded on the left only by padding[1], frequency
    symmetric by padding[0]; two real convs applied to both halves,
    real = rr - ii, imag = ri + ir (biases combine as in the convs)."""
    x = np.pad(x, ((0, 0), (0, 0), (0, 0), (pad_t, 0)))
    r, i = np.split(x, 2, axis=1)
    wr, br = sd[p + 'real_conv.weight'], sd[p + 'real_conv.bias']
    wi, bi = sd[p + 'imag_conv.weight'], sd[p + 'imag_conv.bias']
    conv = lambda a, w, b: nn.conv2d(a, w, b, stride=(2, 1), padding=(pad_f, 0))
    real, imag = _cplx_combine(conv, r, i, wr, br, wi, bi, variant)
    return np.concatenate([real, imag], axis=1)


def _cplx_deconv2d(sd, p, x, variant=0):
    """complexnn.ComplexConvTranspose2d as called at DCCRN_cprs.py:108-115:
    kernel (5,2), stride (2,1), padding (2,0), output_padding (1,0)."""
    r, i = np.split(x, 2, axis=1)
    wr, br = sd[p + 'real_conv.weight'], sd[p + 'real_conv.bias']
    wi, bi = sd[p + 'imag_conv.weight'], sd[p + 'imag_conv.bias']
    dc = lambda a, w, b: nn.conv_transpose2d(a, w, b, stride=(2, 1), padding=(2, 0), output_padding=(1, 0))
    real, imag = _cplx_combine(dc, r, i, wr, br, wi, bi, variant)
    return np.concatenate([real, imag], axis=1)


def _complex_cat(a, b):
    """complexnn.complex_cat([a, b], 1): real halves together, imag halves together."""
    ar, ai = np.split(a, 2, axis=1)
    br, bi = np.split(b, 2, axis=1)
    return np.concatenate([ar, br, ai, bi], axis=1)


def _navie_complex_lstm(sd, p, r, i, proj):
    """complexnn.NavieComplexLSTM.forward([r, i]) as called at DCCRN_cprs.py:182."""
    run = lambda x, q: nn.lstm(x, sd, p + q, 1)
    r2r = run(r, 'real_lstm.')
    r2i = run(r, 'imag_lstm.')
    i2r = run(i, 'real_lstm.')
    i2i = run(i, 'imag_lstm.')
    ro = r2r - i2i
    io = i2r + r2i
    if proj:
        ro = nn.linear(ro, sd[p + 'r_trans.weight'], sd[p + 'r_trans.bias'])
        io = nn.linear(io, sd[p + 'i_trans.weight'], sd[p + 'i_trans.bias'])
    return ro, io


def dccrn_forward(sd, inputs, n_layers=6, masking_mode='E', variant=0):
    """inputs [B,2,F=257,T] -> [B,2,257,T].  DCCRN.forward DCCRN_cprs.py:142-226
    for the decode script's constructor (use_clstm=True, use_cbn=False,
    kernel_num=[32,64,128,256,256,256], rnn_layers=2; dccrn_decode_vb.py:11)."""
    real, imag = inputs[:, 0], inputs[:, -1]                    # :163
    spec_mags = np.sqrt(inputs[:, 0] ** 2 + inputs[:, 1] ** 2)  # torch.norm(dim=1) :164
    spec_phase = np.arctan2(inputs[:, -1], inputs[:, 0])        # :165
    out = inputs[:, :, 1:]                                      # drop DC :166
    enc = []
    for k in range(n_layers):                                   # :170-173
        out = _cplx_conv2d(sd, f'encoder.{k}.0.', out, 2, 1, variant)
        out = nn.prelu(_bn(sd, f'encoder.{k}.1.', out), sd[f'encoder.{k}.2.weight'])
        enc.append(out)
    B, C, D, T = out.shape                                      # :175
    out = np.transpose(out, (3, 0, 1, 2))                       # :176
    r = out[:, :, :C // 2].reshape(T, B, C // 2 * D)            # :178-181
    i = out[:, :, C // 2:].reshape(T, B, C // 2 * D)
    r, i = _navie_complex_lstm(sd, 'enhance.0.', r, i, False)   # :182 (nn.Sequential of 2)
    r, i = _navie_complex_lstm(sd, 'enhance.1.', r, i, True)
    r = r.reshape(T, B, C // 2, D)                              # :183-184
    i = i.reshape(T, B, C // 2, D)
    out = np.concatenate([r, i], axis=2)                        # :185
    out = np.transpose(out, (1, 2, 3, 0))                       # :194
    for k in range(n_layers):                                   # :196-199
        out = np.concatenate([out, enc[-1 - k]], axis=1) if variant & DCCRN_PLAIN_CAT else _complex_cat(out, enc[-1 - k])
        out = _cplx_deconv2d(sd, f'decoder.{k}.0.', out, variant)
        if k < n_layers - 1:
            out = nn.prelu(_bn(sd, f'decoder.{k}.1.', out), sd[f'decoder.{k}.2.weight'])
        out = out[..., 1:]
    mr = np.pad(out[:, 0], ((0, 0), (1, 0), (0, 0)))            # :201-204
    mi = np.pad(out[:, 1], ((0, 0), (1, 0), (0, 0)))
    if masking_mode == 'C':                                     # :220-221 complex ratio mask
        return np.stack([real * mr - imag * mi, real * mi + imag * mr], axis=1)
    if masking_mode == 'R':                                     # :222-223 one real mask per part
        return np.stack([real * mr, imag * mi], axis=1)
    assert masking_mode == 'E'
    mm = (mr ** 2 + mi ** 2) ** 0.5                             # :207
    rp = mr / (mm + 1e-8)
    ip = mi / (mm + 1e-8)
    mph = np.arctan2(ip, rp)                                    # :210-213
    mm = np.tanh(mm)                                            # :216
    em = mm * spec_mags
    ep = spec_phase + mph
    return np.stack([em * np.cos(ep), em * np.sin(ep)], axis=1)  # :219-225


# ----------------------------------------------------------------------------
# FullSubNet   (reference FullSubNet/fullsubnet_net_sa/model.py:68-118, base_model.py:13-42,197-209,
# sequence_model.py:66-84) for the decode script's constructor (fullsubnet_sa_decode_vb.py:11-24).
# ALWAYS batch-1 semantics: `drop_band` (model.py:101-104) runs whenever batch_size > 1 even in eval(), which is a
# training-time trick and changes the output shape; the engine computes B independent batch-1 results instead.
# ----------------------------------------------------------------------------
def _fsn_unfold(x, nn_):
    """BaseModel.unfold: x [B,1,F,T] -> [B,F,2n+1,T] (reflect pad along F, sliding windows)."""
    B, _, F, T = x.shape
    if nn_ < 1:
        return np.transpose(x, (0, 2, 1, 3))                        # [B,F,1,T]
    xp = np.pad(x[:, 0], ((0, 0), (nn_, nn_), (0, 0)), mode='reflect')   # [B,F+2n,T]
    idx = np.arange(F)[:, None] + np.arange(2 * nn_ + 1)[None, :]        # [F, 2n+1]
    return xp[:, idx, :]                                                  # [B,F,2n+1,T]


def _fsn_seq(sd, p, x, act):
    """SequenceModel.forward: x [B,F,T] -> [B,O,T] (LSTM x2 | GRU x2 -> Linear -> act); the cell is read off the state
    dict (nn.GRU stores [3H, .] matrices, nn.LSTM [4H, .]; sequence_model.py:28-43)."""
    H = sd[p + 'sequence_model.weight_hh_l0'].shape[1]
    if sd[p + 'sequence_model.weight_hh_l0'].shape[0] == 3 * H:
        o = nn.gru(np.swapaxes(x, 1, 2), sd, p + 'sequence_model.', 2, batch_first=True)
    else:
        o = nn.lstm(np.swapaxes(x, 1, 2), sd, p + 'sequence_model.', 2, batch_first=True)
    o = nn.linear(o, sd[p + 'fc_output_layer.weight'], sd[p + 'fc_output_layer.bias'])
    if act == 'ReLU':
        o = nn.relu(o)
    return np.swapaxes(o, 1, 2)


def _fsn_cumulative_laplace_norm(a):
    """BaseModel.cumulative_laplace_norm (base_model.py:212-240): a [B,C,F,T] / (mean over (F, frames <= t) per (b, c) + EPSILON);
    float32 sums in the reference's order (sum over F, then a running sum over t)."""
    B, C, F, T = a.shape
    x = a.reshape(B * C, F, T).astype(np.float32)
    step = x.sum(axis=1, dtype=np.float32)                                               # :223
    cum = np.cumsum(step, axis=-1, dtype=np.float32)                                     # :224
    count = np.arange(F, F * T + 1, F, dtype=np.float32)[None, :]                        # :226-233
    mean = (cum / count)[:, None, :]                                                     # :235-236
    return (x / (mean + np.finfo(np.float32).eps)).reshape(B, C, F, T)                   # :238 (constant.py:8 EPSILON)


def fullsubnet_forward(sd, noisy_mag, look_ahead=2, sb_nn=15, fb_nn=0, norm_type='offline_laplace_norm'):
    """noisy_mag [B,1,257,T] -> complex mask [B,2,257,T]; every utterance processed with batch-1 semantics.
    norm_type: 'offline_laplace_norm' (the decode script's, base_model.py:197-209) or 'cumulative_laplace_norm' (:212-240, the
    causal one: with it the network only looks `look_ahead` frames ahead)."""
    outs = []
    for b in range(noisy_mag.shape[0]):
        x = np.pad(noisy_mag[b:b + 1], ((0, 0), (0, 0), (0, 0), (0, look_ahead)))          # :79
        _, _, F, T = x.shape
        if norm_type == 'cumulative_laplace_norm':
            norm = _fsn_cumulative_laplace_norm
        else:
            norm = lambda a: a / (a.mean(axis=(1, 2, 3), keepdims=True) + 1e-5)           # offline_laplace_norm
        fb_in = norm(x).reshape(1, F, T)                                                  # :84
        fb_out = _fsn_seq(sd, 'fb_model.', fb_in, 'ReLU').reshape(1, 1, F, T)             # :85
        fb_unf = _fsn_unfold(fb_out, fb_nn).reshape(1, F, 2 * fb_nn + 1, T)               # :88-89
        nz_unf = _fsn_unfold(x, sb_nn).reshape(1, F, 2 * sb_nn + 1, T)                    # :92-93
        sb_in = norm(np.concatenate([nz_unf, fb_unf], axis=2))                            # :96-97
        sb_in = sb_in.reshape(F, 2 * sb_nn + 1 + 2 * fb_nn + 1, T)                        # :106-110
        m = _fsn_seq(sd, 'sb_model.', sb_in, None)                                        # :113
        m = np.transpose(m.reshape(1, F, 2, T), (0, 2, 1, 3))                             # :114
        outs.append(m[:, :, :, look_ahead:])                                              # :117
    return np.concatenate(outs, axis=0)


# ----------------------------------------------------------------------------
# GCRN   (reference GCRN/GCRN_noncprs.py:5-165)
# ----------------------------------------------------------------------------
def _glu_conv(sd, p, x):
    """GluConv2d (GCRN_noncprs.py:42-60): conv1(x) * sigmoid(conv2(x)), kernel (1,3), stride (1,2)."""
    a = nn.conv2d(x, sd[p + 'conv1.weight'], sd[p + 'conv1.bias'], stride=(1, 2))
    g = nn.conv2d(x, sd[p + 'conv2.weight'], sd[p + 'conv2.bias'], stride=(1, 2))
    return a * nn.sigmoid(g)


def _glu_deconv(sd, p, x, out_pad=0):
    """GluConvTranspose2d (:63-83)."""
    a = nn.conv_transpose2d(x, sd[p + 'conv1.weight'], sd[p + 'conv1.bias'], stride=(1, 2), output_padding=(0, out_pad))
    g = nn.conv_transpose2d(x, sd[p + 'conv2.weight'], sd[p + 'conv2.bias'], stride=(1, 2), output_padding=(0, out_pad))
    return a * nn.sigmoid(g)


def _glstm(sd, x):
    """GLSTM.forward (:22-39).  x [B,256,T,4]."""
    B, C, T, F = x.shape
    out = np.transpose(x, (0, 2, 1, 3)).reshape(B, T, -1)
    ch = np.split(out, 2, axis=-1)
    hs = [nn.lstm(ch[i], sd, f'glstm.lstm_list1.{i}.', 1, batch_first=True) for i in range(2)]
    out = np.stack(hs, axis=-1).reshape(B, T, -1)                # stack on a new last dim, then flatten: interleave
    out = nn.layernorm(out, sd['glstm.ln1.weight'], sd['glstm.ln1.bias'], 1)
    ch = np.split(out, 2, axis=-1)
    out = np.concatenate([nn.lstm(ch[i], sd, f'glstm.lstm_list2.{i}.', 1, batch_first=True) for i in range(2)], axis=-1)
    out = nn.layernorm(out, sd['glstm.ln2.weight'], sd['glstm.ln2.bias'], 1)
    return np.transpose(out.reshape(B, T, C, F), (0, 2, 1, 3))


def gcrn_forward(sd, x):
    """x [B,2,T,161] (RI) -> [B,2,T,161] (RI mapping).  Net.forward GCRN_noncprs.py:135-165."""
    e = []
    out = x
    for k in range(1, 6):                                       # :137-141
        out = nn.elu(_bn(sd, f'bn{k}.', _glu_conv(sd, f'conv{k}.', out)))
        e.append(out)
    out = _glstm(sd, out)                                       # :145
    out = np.concatenate([out, e[4]], axis=1)                   # :147
    res = []
    for br in (1, 2):                                           # :149-159
        d = out
        for k in (5, 4, 3, 2):
            y = _bn(sd, f'bn{k}_t_{br}.', _glu_deconv(sd, f'conv{k}_t_{br}.', d, 1 if k == 2 else 0))
            d = nn.elu(np.concatenate([y, e[k - 2]], axis=1))
        d = nn.elu(_bn(sd, f'bn1_t_{br}.', _glu_deconv(sd, f'conv1_t_{br}.', d)))
        res.append(nn.linear(d, sd[f'fc{br}.weight'], sd[f'fc{br}.bias']))   # :161-162 (over the F axis)
    return np.concatenate(res, axis=1)                          # :163


# ----------------------------------------------------------------------------
# CTSNet   (reference CTSNet/Step1_network.py:12-211, CTSNet/Step2_network.py:13-210)
# ----------------------------------------------------------------------------
def _norm(sd, p, x):
    """InstanceNorm{1,2}d(affine) of the base models, or the CumulativeLayerNorm of the `_new` variants (same position
    in every nn.Sequential; parameters are `gain` / `bias` there) - selected by the keys present in the state dict."""
    if p + 'gain' in sd:
        return nn.cumulative_layernorm(x, sd[p + 'gain'], sd[p + 'bias'])
    return nn.instancenorm(x, sd[p + 'weight'], sd[p + 'bias'])


def _in2d(sd, p, x):
    return _norm(sd, p, x)


def _cts_gate_conv(sd, p, x, stride=(1, 2)):
    """Gate_Conv de_flag=0 (Step1_network.py:121-133): pad one frame on top, conv * sigmoid(gate_conv)."""
    x = np.pad(x, ((0, 0), (0, 0), (1, 0), (0, 0)))
    a = nn.conv2d(x, sd[p + 'conv.1.weight'], sd[p + 'conv.1.bias'], stride=stride)
    g = nn.conv2d(x, sd[p + 'gate_conv.1.weight'], sd[p + 'gate_conv.1.bias'], stride=stride)
    return a * nn.sigmoid(g)


def _cts_gate_deconv(sd, p, x):
    """Gate_Conv de_flag=1 (:134-145): ConvTranspose2d + Chomp_T(1) on both branches."""
    a = nn.conv_transpose2d(x, sd[p + 'conv.0.weight'], sd[p + 'conv.0.bias'], stride=(1, 2))[:, :, :-1]
    g = nn.conv_transpose2d(x, sd[p + 'gate_conv.0.weight'], sd[p + 'gate_conv.0.bias'], stride=(1, 2))[:, :, :-1]
    return a * nn.sigmoid(g)


def _share_sep_conv(w, x):
    """ShareSepConv (:190-204): one FIR shared by all channels, causal left pad K-1.  x [B,C,T], w [1,1,K]."""
    K = w.shape[-1]
    xp = np.pad(x, ((0, 0), (0, 0), (K - 1, 0)))
    out = np.zeros_like(x)
    for k in range(K):
        out += w[0, 0, k].astype(x.dtype) * xp[:, :, k:k + x.shape[-1]]
    return out


def _cts_glu(sd, p, x, d, left='left_conv', right='right_conv'):
    """Glu / glu block (Step1_network.py:158-188, Step2_network.py:126-158).  x [B,256,T]."""
    resi = x
    x = nn.conv1d(x, sd[p + 'in_conv.weight'])

    def branch(name):
        y = nn.prelu(x, sd[p + name + '.0.weight'])
        y = _norm(sd, p + name + '.1.', y)
        y = _share_sep_conv(sd[p + name + '.2.weight'], y)
        y = np.pad(y, ((0, 0), (0, 0), (4 * d, 0)))
        return nn.conv1d(y, sd[p + name + '.4.weight'], dilation=d)
    x = branch(left) * nn.sigmoid(branch(right))
    y = nn.prelu(x, sd[p + 'out_conv.0.weight'])
    y = _norm(sd, p + 'out_conv.1.', y)
    return nn.conv1d(y, sd[p + 'out_conv.2.weight']) + resi


def _cts_encoder(sd, p, x):
    skips = []
    for i in range(5):
        x = _cts_gate_conv(sd, f'{p}{i}.0.', x)
        x = nn.prelu(_in2d(sd, f'{p}{i}.1.', x), sd[f'{p}{i}.2.weight'])
        skips.append(x)
    return x, skips


def _cts_decoder(sd, p, p6, x, skips, softplus):
    for i in range(5):
        x = np.concatenate([x, skips[-(i + 1)]], axis=1)
        x = _cts_gate_deconv(sd, f'{p}{i}.0.', x)
        x = nn.prelu(_in2d(sd, f'{p}{i}.1.', x), sd[f'{p}{i}.2.weight'])
    x = nn.linear(x[:, 0], sd[p6 + '0.weight'], sd[p6 + '0.bias'])
    return nn.softplus(x) if softplus else x


def cts_step1_forward(sd, x):
    """Step1_net.forward (Step1_network.py:21-40): magnitude [B,T,161] -> magnitude [B,T,161]."""
    x, skips = _cts_encoder(sd, 'en.en.', x[:, None])
    B, _, T, _ = x.shape
    x = np.transpose(x, (0, 1, 3, 2)).reshape(B, -1, T)
    acc = np.zeros_like(x)
    for k in (1, 2, 3):
        for i in range(6):
            x = _cts_glu(sd, f'tcm{k}.tcm_list.{i}.', x, 2 ** i)
        acc = acc + x
    x = np.transpose(acc.reshape(B, 64, 4, T), (0, 1, 3, 2))
    return _cts_decoder(sd, 'de.de.', 'de.de6.', x, skips, True)


def cts_step2_forward(sd, inpt, R=None, X=None):
    """Step2_net.forward (Step2_network.py:23-38): [B,4,T,161] -> [B,2,T,161].  R groups of X gated blocks (:13-21): as many
    as the state dict holds (3 x 6 in the decode script, two_stage_com_decode_vb.py:14)."""
    if R is None:
        R = _module_list_len(sd, 'tcm_list.')
    if X is None:
        X = _module_list_len(sd, 'tcm_list.0.glu_list.')
    x, skips = _cts_encoder(sd, 'en.en_module.', inpt)
    B, _, T, _ = x.shape
    x = np.transpose(x, (0, 1, 3, 2)).reshape(B, -1, T)
    acc = np.zeros_like(x)
    for r in range(R):
        for i in range(X):
            x = _cts_glu(sd, f'tcm_list.{r}.glu_list.{i}.', x, 2 ** i, 'ori_conv', 'att_ori')
        acc = acc + x
    x = np.transpose(acc.reshape(B, 64, 4, T), (0, 1, 3, 2))
    xr = _cts_decoder(sd, 'de_r.de_list.', 'de_r.de6.', x, skips, False)
    xi = _cts_decoder(sd, 'de_i.de_list.', 'de_i.de6.', x, skips, False)
    return np.stack([xr, xi], axis=1)


# ----------------------------------------------------------------------------
# TaylorSENet   (reference TaylorSENet/TaylorSENet.py:8-693) for the decode script's constructor
# (taylorsenet_decode_vb.py:11-13): k1=(1,3), k2=(2,3), c=64, kd1=5, cd1=64, d_feat=256, dilations [1,2,5,9], p=2,
# order_num=3, intra/inter 'cat', causal, no conformer, U2 encoder, no sharing.
# ----------------------------------------------------------------------------
def _in_prelu(sd, p_in, p_pr, x):
    return nn.prelu(_norm(sd, p_in, x), sd[p_pr + 'weight'])


def _gate_conv2d(sd, p, x, kt):
    """GateConv2d (TaylorSENet.py:549-575): ONE conv to 2C channels, out * sigmoid(gate); top pad kt-1 frames."""
    key = p + ('conv.1.' if kt > 1 else 'conv.')
    if kt > 1:
        x = np.pad(x, ((0, 0), (0, 0), (kt - 1, 0), (0, 0)))
    y = nn.conv2d(x, sd[key + 'weight'], sd[key + 'bias'], stride=(1, 2))
    a, g = np.split(y, 2, axis=1)
    return a * nn.sigmoid(g)


def _gate_deconv2d(sd, p, x, kt):
    """GateConvTranspose2d (:577-603) with Chomp_T(kt-1)."""
    key = p + ('conv.0.' if kt > 1 else 'conv.')
    y = nn.conv_transpose2d(x, sd[key + 'weight'], sd[key + 'bias'], stride=(1, 2))
    if kt > 1:
        y = y[:, :, :-(kt - 1)]
    a, g = np.split(y, 2, axis=1)
    return a * nn.sigmoid(g)


def _en_unet_module(sd, p, x, k1t, scale, de_flag):
    """En_unet_module.forward (:480-496)."""
    y = _gate_deconv2d(sd, p + 'in_conv.0.', x, k1t) if de_flag else _gate_conv2d(sd, p + 'in_conv.0.', x, k1t)
    x_resi = _in_prelu(sd, p + 'in_conv.1.', p + 'in_conv.2.', y)
    x = x_resi
    xs = []
    for i in range(scale):                                       # Conv2dunit k2=(2,3) (:498-519)
        q = f'{p}enco.{i}.conv.'
        x = nn.conv2d(np.pad(x, ((0, 0), (0, 0), (1, 0), (0, 0))), sd[q + '1.weight'], sd[q + '1.bias'], stride=(1, 2))
        x = _in_prelu(sd, q + '2.', q + '3.', x)
        xs.append(x)
    for i in range(scale):                                       # Deconv2dunit (:521-547)
        q = f'{p}deco.{i}.deconv.'
        if i > 0:
            x = np.concatenate([x, xs[-(i + 1)]], axis=1)
        x = nn.conv_transpose2d(x, sd[q + '0.weight'], sd[q + '0.bias'], stride=(1, 2))[:, :, :-1]
        x = _in_prelu(sd, q + '2.', q + '3.', x)
    return x_resi + x


def _u2net_encoder(sd, p, x):
    """U2Net_Encoder.forward (:363-370)."""
    ens = []
    for i, (k1t, scale) in enumerate(((2, 4), (1, 3), (1, 2), (1, 1))):
        x = _en_unet_module(sd, f'{p}meta_unet_list.{i}.', x, k1t, scale, False)
        ens.append(x)
    x = _in_prelu(sd, p + 'last_conv.1.', p + 'last_conv.2.', _gate_conv2d(sd, p + 'last_conv.0.', x, 1))
    ens.append(x)
    return x, ens


def _u2net_decoder(sd, p, x, ens):
    """U2Net_Decoder.forward, inter_connect='cat' (:426-439)."""
    for i in range(4):
        x = _en_unet_module(sd, f'{p}meta_unet_list.{i}.', np.concatenate([x, ens[-(i + 1)]], axis=1), 1, i + 1, True)
    x = np.concatenate([x, ens[0]], axis=1)
    x = _in_prelu(sd, p + 'last_conv.1.', p + 'last_conv.2.', _gate_deconv2d(sd, p + 'last_conv.0.', x, 2))
    x = nn.sigmoid(nn.conv2d(x, sd[p + 'last_conv.3.weight'], sd[p + 'last_conv.3.bias']))
    return x[:, 0]


def _squeezed_tcm(sd, p, x, d, k=5):
    """SqueezedTCM.forward (:679-685), causal pad (k-1)*d."""
    resi = x
    x = nn.conv1d(x, sd[p + 'in_conv.weight'])

    def branch(name):
        y = nn.prelu(x, sd[p + name + '.0.weight'])
        y = _norm(sd, p + name + '.1.', y)
        y = np.pad(y, ((0, 0), (0, 0), ((k - 1) * d, 0)))
        return nn.conv1d(y, sd[p + name + '.3.weight'], dilation=d)
    x = branch('left_conv') * nn.sigmoid(branch('right_conv'))
    y = _norm(sd, p + 'out_conv.1.', nn.prelu(x, sd[p + 'out_conv.0.weight']))
    return nn.conv1d(y, sd[p + 'out_conv.2.weight']) + resi


def _tcms(sd, p, x, n=2, dils=(1, 2, 5, 9)):
    for i in range(n):
        for j, d in enumerate(dils):
            x = _squeezed_tcm(sd, f'{p}tcms.{i}.tcm_list.{j}.', x, d)
    return x


def _module_list_len(sd, prefix):
    """len() of the ModuleList whose entries are `prefix<i>.` in a state dict (0 if absent)."""
    return 1 + max((int(k[len(prefix):].split('.', 1)[0]) for k in sd if k.startswith(prefix)), default=-1)


def taylorsenet_forward(sd, inputs, order_num=None):
    """TaylorSENet.forward (:66-94): [B,2,T,161] -> [B,2,T,161].  order_num (:27,66-70): as many high-order blocks as
    the state dict holds (3 in the decode script, taylorsenet_decode_vb.py:11-13)."""
    if order_num is None:
        order_num = _module_list_len(sd, 'highorderblock_list.')
    mag = np.sqrt(inputs[:, 0] ** 2 + inputs[:, 1] ** 2)
    ph = np.arctan2(inputs[:, -1], inputs[:, 0])
    # ZeroOrderBlock (:139-153)
    en_x, ens = _u2net_encoder(sd, 'zeroorderblock.en.', inputs)
    B, C, T, F = en_x.shape
    x = _tcms(sd, 'zeroorderblock.', np.swapaxes(en_x, -2, -1).reshape(B, C * F, T))
    gain = _u2net_decoder(sd, 'zeroorderblock.de.', np.swapaxes(x.reshape(B, C, F, T), -2, -1), ens)
    zmag = gain * mag
    zero = np.stack([zmag * np.cos(ph), zmag * np.sin(ph)], axis=1)          # :73-76
    fh, _ = _u2net_encoder(sd, 'separate_en.', inputs)                       # :78-82
    fh = np.swapaxes(fh, -2, -1).reshape(B, -1, T)
    out, pre = zero, zero
    fact = 1.0
    for k in range(order_num):                                               # :84-93
        p = f'highorderblock_list.{k}.'
        x1 = np.swapaxes(pre, -2, -1).reshape(B, -1, T)                      # HighOrderBlock.forward :191-214
        x = nn.conv1d(np.concatenate([fh, x1], axis=1), sd[p + 'in_conv.weight'], sd[p + 'in_conv.bias'])
        x = _tcms(sd, p, x)
        xr = np.swapaxes(nn.conv1d(x, sd[p + 'real_resi.weight'], sd[p + 'real_resi.bias']), -2, -1)
        xi = np.swapaxes(nn.conv1d(x, sd[p + 'imag_resi.weight'], sd[p + 'imag_resi.bias']), -2, -1)
        upd = np.stack([xr, xi], axis=1) + k * pre
        pre = upd
        fact *= (k + 1)
        out = out + upd / fact
    return out


# ----------------------------------------------------------------------------
# G2Net   (reference G2Net_VB/gaf_net_320.py:10-526) for the decode script's constructor (com_decode.py:23):
# gaf_base(3, 64, 2, 4, 4, [1,2,5,9], 256+161*2, 256, 256, (2,3), (1,3), 64, 'cat', 3, is_aux=False,
#          encoder_type='U2Net', tcm_type='full-band')
# ----------------------------------------------------------------------------
def _g2_gate_conv(sd, p, x):
    """Gate_2dconv de_flag=False (gaf_net_320.py:465-486): two convs, top pad 1."""
    x = np.pad(x, ((0, 0), (0, 0), (1, 0), (0, 0)))
    a = nn.conv2d(x, sd[p + 'conv.1.weight'], sd[p + 'conv.1.bias'], stride=(1, 2))
    g = nn.conv2d(x, sd[p + 'gate_conv.1.weight'], sd[p + 'gate_conv.1.bias'], stride=(1, 2))
    return a * nn.sigmoid(g)


def _g2_unet_module(sd, p, x, scale):
    """En_unet_module.forward (:415-431), k2 = (1,3): inner (de)convs have no time extent."""
    x_resi = _in_prelu(sd, p + 'in_conv.1.', p + 'in_conv.2.', _g2_gate_conv(sd, p + 'in_conv.0.', x))
    x = x_resi
    xs = []
    for i in range(scale):
        q = f'{p}enco.{i}.conv.'
        x = _in_prelu(sd, q + '1.', q + '2.', nn.conv2d(x, sd[q + '0.weight'], sd[q + '0.bias'], stride=(1, 2)))
        xs.append(x)
    for i in range(scale):
        q = f'{p}deco.{i}.deconv.'
        if i > 0:
            x = np.concatenate([x, xs[-(i + 1)]], axis=1)
        x = _in_prelu(sd, q + '1.', q + '2.', nn.conv_transpose2d(x, sd[q + '0.weight'], sd[q + '0.bias'], stride=(1, 2)))
    return x_resi + x


def _g2_glu(sd, p, x, d):
    """Glu (:245-274): single (un-gated) branch, k = 3, causal."""
    resi = x
    x = nn.conv1d(x, sd[p + 'in_conv.weight'])
    y = _norm(sd, p + 'left_conv.1.', nn.prelu(x, sd[p + 'left_conv.0.weight']))
    x = nn.conv1d(np.pad(y, ((0, 0), (0, 0), (2 * d, 0))), sd[p + 'left_conv.3.weight'], dilation=d)
    y = _norm(sd, p + 'out_conv.1.', nn.prelu(x, sd[p + 'out_conv.0.weight']))
    return nn.conv1d(y, sd[p + 'out_conv.2.weight']) + resi


def _g2_tcm_seq(sd, p, x, n_out_idx=2, dils=(1, 2, 5, 9)):
    """nn.Sequential(*[Tcm_list]*2, Conv1d(256,161,1)[, Sigmoid])."""
    for i in range(2):
        for j, d in enumerate(dils):
            x = _g2_glu(sd, f'{p}{i}.tcm_list.{j}.', x, d)
    return nn.conv1d(x, sd[f'{p}{n_out_idx}.weight'], sd[f'{p}{n_out_idx}.bias'])


def g2net_forward(sd, inpt, stage_num=None):
    """gaf_base.forward (:73-87): [B,2,T,161] -> list of stage outputs [B,2,161,T].  stage_num (:27,55-58): as many GAF
    stages as the state dict holds (3 in the decode script, com_decode.py:23)."""
    if stage_num is None:
        stage_num = _module_list_len(sd, 'gafs.')
    B, _, T, _ = inpt.shape
    x = inpt
    for i, scale in enumerate((4, 3, 2, 1)):                     # U2Net_Encoder (:277-304)
        x = _g2_unet_module(sd, f'en.meta_unet_list.{i}.', x, scale)
    x = _in_prelu(sd, 'en.last_conv.1.', 'en.last_conv.2.', _g2_gate_conv(sd, 'en.last_conv.0.', x))
    feat = np.swapaxes(x, -2, -1).reshape(B, -1, T)
    pre = np.swapaxes(inpt, -2, -1)                               # [B,2,161,T]
    outs = []
    for s in range(stage_num):                                    # GAF_module.forward (:104-115)
        p = f'gafs.{s}.'
        pmag = np.sqrt(pre[:, 0] ** 2 + pre[:, 1] ** 2)
        pph = np.arctan2(pre[:, -1], pre[:, 0])
        xin = np.concatenate([feat, pre.reshape(B, -1, T)], axis=1)

        def gate_in(q):
            a = nn.conv1d(xin, sd[q + 'in_conv_main.weight'], sd[q + 'in_conv_main.bias'])
            g = nn.conv1d(xin, sd[q + 'in_conv_gate.0.weight'], sd[q + 'in_conv_gate.0.bias'])
            return a * nn.sigmoid(g)
        gain = nn.sigmoid(_g2_tcm_seq(sd, p + 'glance_branch.mstcm_filter.', gate_in(p + 'glance_branch.')))   # :142-146
        xf = gate_in(p + 'focus_branch.')                                                                          # :179-183
        resi = np.stack([_g2_tcm_seq(sd, p + 'focus_branch.mstcm_r.', xf), _g2_tcm_seq(sd, p + 'focus_branch.mstcm_i.', xf)], 1)
        xm = pmag * gain
        pre = np.stack([xm * np.cos(pph), xm * np.sin(pph)], 1) + resi
        outs.append(pre)
    return outs


# ----------------------------------------------------------------------------
# Uformer   (reference Uformer/uformer.py:30-287 + conv2d_cplx.py, conv2d_real.py, fusion.py,
# dilated_dualpath_conformer.py, ff_cplx.py, ff_real.py, linear_cplx.py, linear_real.py, t_att_cplx.py, f_att_cplx.py,
# t_att_real.py, f_att_real.py, dsconv2d_cplx.py, dsconv2d_real.py).  Complex tensors are (real, imag) pairs of
# [N,C,F,T] arrays (the reference stacks them on a trailing axis of size 2).
# ----------------------------------------------------------------------------
_UEPS = float(np.finfo(np.float32).eps)       # uformer.py:16 EPSILON


def _u_ln(sd, p, x):
    """nn.LayerNorm(C) applied over the channel axis of [N,C,F,T] (x.transpose(1,-1) in the reference)."""
    mu = x.mean(axis=1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=1, keepdims=True)
    w = sd[p + 'weight'].astype(x.dtype).reshape(1, -1, 1, 1)
    b = sd[p + 'bias'].astype(x.dtype).reshape(1, -1, 1, 1)
    return (x - mu) / np.sqrt(var + 1e-5) * w + b


def _u_lin(w, b, x):
    return np.einsum('oc,ncft->noft', w.astype(x.dtype), x, optimize=True) + b.astype(x.dtype).reshape(1, -1, 1, 1)


def _u_rlin(sd, p, x):                         # Real_Linear (linear_real.py:11-24)
    return _u_lin(sd[p + 'linear.weight'], sd[p + 'linear.bias'], x)


def _u_clin(sd, p, r, i):                      # Complex_Linear (linear_cplx.py:11-27)
    wr, br, wi, bi = sd[p + 'real_linear.weight'], sd[p + 'real_linear.bias'], sd[p + 'imag_linear.weight'], sd[p + 'imag_linear.bias']
    return _u_lin(wr, br, r) - _u_lin(wi, bi, i), _u_lin(wr, br, i) + _u_lin(wi, bi, r)


def _u_fusion(r, i, mag):                      # fusion.py:13-19
    cm = np.sqrt(np.maximum(r ** 2 + i ** 2, _UEPS))
    s = nn.sigmoid(mag)
    return r + s, i + s, mag + nn.sigmoid(cm)


def _u_att(q, k, v, axis):
    """softmax(Q K^T / 4) V along `axis` (3 = time, 2 = frequency) of [N,16,F,T]."""
    if axis == 3:
        e = np.einsum('ndft,ndfs->nfts', q, k, optimize=True) / 4.0
    else:
        e = np.einsum('ndft,ndgt->ntfg', q, k, optimize=True) / 4.0
    e = e - e.max(axis=-1, keepdims=True)
    pr = np.exp(e)
    pr = pr / pr.sum(axis=-1, keepdims=True)
    if axis == 3:
        return np.einsum('nfts,ndfs->ndft', pr, v, optimize=True)
    return np.einsum('ntfg,ndgt->ndft', pr, v, optimize=True)


def _u_catt(sd, p, r, i, axis, nm):
    """Multihead_Attention_{T,F}_Branch (t_att_cplx.py:73-96 / f_att_cplx.py:65-88), one head."""
    h = p + 'attn_heads.0.'
    xr, xi = _u_ln(sd, h + 'layernorm1.', r), _u_ln(sd, h + 'layernorm1.', i)
    src = {'r': xr, 'i': xi}
    combos = ['rrr', 'rii', 'iri', 'iir', 'rri', 'rir', 'irr', 'iii']        # (q, k, v) sources of att1..att8
    outs = []
    for n, cmb in enumerate(combos):
        a = f'{h}{nm}_att{n + 1}.'
        outs.append(_u_att(_u_rlin(sd, a + 'query.', src[cmb[0]]), _u_rlin(sd, a + 'key.', src[cmb[1]]),
                           _u_rlin(sd, a + 'value.', src[cmb[2]]), axis))
    A, B, C, D, E, F_, G, H = outs
    ar, ai = A - B - C - D, E + F_ + G - H
    ar, ai = _u_ln(sd, h + 'layernorm2.', ar), _u_ln(sd, h + 'layernorm2.', ai)
    tr, ti = _u_clin(sd, p + 'transform_linear.', ar, ai)
    pw = sd[p + 'prelu.weight']
    return nn.prelu(_u_ln(sd, p + 'layernorm3.', tr), pw) + r, nn.prelu(_u_ln(sd, p + 'layernorm3.', ti), pw) + i


def _u_ratt(sd, p, x, axis, nm):
    """Multihead_Attention_{T,F}_Branch_real (t_att_real.py:53-76)."""
    h = p + 'attn_heads.0.'
    y = _u_ln(sd, h + 'layernorm1.', x)
    a = f'{h}{nm}_att.'
    y = _u_att(_u_rlin(sd, a + 'query.', y), _u_rlin(sd, a + 'key.', y), _u_rlin(sd, a + 'value.', y), axis)
    y = _u_ln(sd, h + 'layernorm2.', y)
    y = _u_rlin(sd, p + 'transform_linear.', y)
    return nn.prelu(_u_ln(sd, p + 'layernorm3.', y), sd[p + 'prelu.weight']) + x


def _u_cconv(sd, p, r, i, stride, pad, dil, T, transposed=False, out_pad=(0, 0)):
    """ComplexConv2d_Encoder / _Decoder (conv2d_cplx.py:11-68): two real (de)convs, output cut to the first T frames."""
    wr, br, wi, bi = sd[p + 'real_conv.weight'], sd[p + 'real_conv.bias'], sd[p + 'imag_conv.weight'], sd[p + 'imag_conv.bias']
    if transposed:
        f = lambda a, w, b: nn.conv_transpose2d(a, w, b, stride=stride, padding=pad, output_padding=out_pad)
    else:
        f = lambda a, w, b: nn.conv2d(a, w, b, stride=stride, padding=pad, dilation=dil)
    return (f(r, wr, br) - f(i, wi, bi))[..., :T], (f(i, wr, br) + f(r, wi, bi))[..., :T]


def _u_rconv(sd, p, x, stride, pad, dil, T, transposed=False, out_pad=(0, 0)):
    if transposed:
        return nn.conv_transpose2d(x, sd[p + 'conv.weight'], sd[p + 'conv.bias'], stride=stride, padding=pad, output_padding=out_pad)[..., :T]
    return nn.conv2d(x, sd[p + 'conv.weight'], sd[p + 'conv.bias'], stride=stride, padding=pad, dilation=dil)[..., :T]


def _u_ff(sd, p, r, i=None):
    """FF_Cplx (ff_cplx.py:10-33) / FF_Real (ff_real.py:11-33): LN -> Linear -> PReLU -> Linear, y*0.5 + x."""
    pw = sd[p + 'prelu.weight']
    if i is None:
        y = _u_rlin(sd, p + 'linear1.', _u_ln(sd, p + 'layernorm_linear.', r))
        return _u_rlin(sd, p + 'linear2.', nn.prelu(y, pw)) * 0.5 + r
    yr, yi = _u_clin(sd, p + 'linear1.', _u_ln(sd, p + 'layernorm_linear.', r), _u_ln(sd, p + 'layernorm_linear.', i))
    yr, yi = _u_clin(sd, p + 'linear2.', nn.prelu(yr, pw), nn.prelu(yi, pw))
    return yr * 0.5 + r, yi * 0.5 + i


def _u_dsconv(sd, p, d1, d2, r, i=None):
    """DSConv2d (dsconv2d_cplx.py:44-60) / DSConv2d_Real (dsconv2d_real.py:44-60)."""
    T = r.shape[-1]
    pw = sd[p + 'prelu.weight']
    if i is None:
        y = nn.prelu(_u_rconv(sd, p + 'conv1x1.', _u_ln(sd, p + 'layernorm_conv1.', r), 1, 0, 1, T), pw)
        y1 = _u_rconv(sd, p + 'dconv1.', y, 1, (1, d1), (1, d1), T)
        y2 = _u_rconv(sd, p + 'dconv2.', y, 1, (1, d2), (1, d2), T)
        y = _u_ln(sd, p + 'layernorm_conv2.', y1 * nn.sigmoid(y2))
        return r + _u_rconv(sd, p + 'sconv.', y * nn.sigmoid(y), 1, 0, 1, T)
    yr, yi = _u_cconv(sd, p + 'conv1x1.', _u_ln(sd, p + 'layernorm_conv1.', r), _u_ln(sd, p + 'layernorm_conv1.', i), 1, 0, 1, T)
    yr, yi = nn.prelu(yr, pw), nn.prelu(yi, pw)
    ar, ai = _u_cconv(sd, p + 'dconv1.', yr, yi, 1, (1, d1), (1, d1), T)
    br, bi = _u_cconv(sd, p + 'dconv2.', yr, yi, 1, (1, d2), (1, d2), T)
    yr, yi = _u_ln(sd, p + 'layernorm_conv2.', ar * nn.sigmoid(br)), _u_ln(sd, p + 'layernorm_conv2.', ai * nn.sigmoid(bi))
    sr, si = _u_cconv(sd, p + 'sconv.', yr * nn.sigmoid(yr), yi * nn.sigmoid(yi), 1, 0, 1, T)
    return r + sr, i + si


def uformer_core(sd, re, im):
    """Spectral part of Uformer.forward (uformer.py:197-262): noisy STFT (re, im) [N,257,T] ->
    enhanced (real, imag) [N,257,T].  The STFT / iSTFT around it are in oracle/decode.py."""
    T = re.shape[-1]
    mag = np.sqrt(np.maximum(re ** 2 + im ** 2, _UEPS))                     # :197
    phase = np.arctan2(im + _UEPS, re)
    mag0 = mag[:, None]
    r, i = (mag * np.cos(phase))[:, None, 1:], (mag * np.sin(phase))[:, None, 1:]   # :205-210
    m = mag0[:, :, 1:]
    enc, menc = [], []
    for k in range(6):                                                      # :214-219
        r, i = _u_cconv(sd, f'encoder.{k}.0.', r, i, (2, 1), (2, 1), 1, T)
        bn = lambda a: nn.batchnorm(a, sd[f'encoder.{k}.1.weight'], sd[f'encoder.{k}.1.bias'],
                                    sd[f'encoder.{k}.1.running_mean'], sd[f'encoder.{k}.1.running_var'])
        r, i = nn.prelu(bn(r), sd[f'encoder.{k}.2.weight']), nn.prelu(bn(i), sd[f'encoder.{k}.2.weight'])
        m = _u_rconv(sd, f'encoder_real.{k}.0.', m, (2, 1), (2, 1), 1, T)
        m = nn.prelu(_bn(sd, f'encoder_real.{k}.1.', m), sd[f'encoder_real.{k}.2.weight'])
        r, i, m = _u_fusion(r, i, m)
        enc.append((r, i))
        menc.append(m)
    # Dilated_Dualpath_Conformer.forward (dilated_dualpath_conformer.py:53-78)
    c = 'conformer.'
    r, i = _u_ff(sd, c + 'ff1_cplx.', r, i)
    m = _u_ff(sd, c + 'ff1_mag.', m)
    r, i, m = _u_fusion(r, i, m)
    r, i = _u_catt(sd, c + 'cplx_tatt.', r, i, 3, 'T')
    m = _u_ratt(sd, c + 'mag_tatt.', m, 3, 'T')
    r, i, m = _u_fusion(r, i, m)
    r, i = _u_catt(sd, c + 'cplx_fatt.', r, i, 2, 'F')
    m = _u_ratt(sd, c + 'mag_fatt.', m, 2, 'F')
    r, i, m = _u_fusion(r, i, m)
    dil = [1, 2, 4, 8, 16, 32, 64, 128]
    for k in range(8):
        r, i = _u_dsconv(sd, f'{c}dsconv_cplx.{k}.', dil[k], dil[7 - k], r, i)
        m = _u_dsconv(sd, f'{c}dsconv_real.{k}.', dil[k], dil[7 - k], m)
        r, i, m = _u_fusion(r, i, m)
    r, i = _u_ff(sd, c + 'ff2_cplx.', r, i)
    m = _u_ff(sd, c + 'ff2_mag.', m)
    r, i, m = _u_fusion(r, i, m)
    r, i, m = _u_ln(sd, c + 'ln_conformer_cplx.', r), _u_ln(sd, c + 'ln_conformer_cplx.', i), _u_ln(sd, c + 'ln_conformer_mag.', m)
    for k in range(6):                                                      # :225-232
        er, ei = enc[-1 - k]
        r, i = _u_cconv(sd, f'decoder.{k}.0.', np.concatenate([er, r], 1), np.concatenate([ei, i], 1), (2, 1), (2, 0), 1, T,
                        True, (1, 0))
        m = _u_rconv(sd, f'decoder_real.{k}.0.', np.concatenate([menc[-1 - k], m], 1), (2, 1), (2, 0), 1, T, True, (1, 0))
        if k < 5:
            bn = lambda a: nn.batchnorm(a, sd[f'decoder.{k}.1.weight'], sd[f'decoder.{k}.1.bias'],
                                        sd[f'decoder.{k}.1.running_mean'], sd[f'decoder.{k}.1.running_var'])
            r, i = nn.prelu(bn(r), sd[f'decoder.{k}.2.weight']), nn.prelu(bn(i), sd[f'decoder.{k}.2.weight'])
            m = nn.prelu(_bn(sd, f'decoder_real.{k}.1.', m), sd[f'decoder_real.{k}.2.weight'])
        r, i, m = _u_fusion(r, i, m)
    m = np.pad(nn.sigmoid(m), ((0, 0), (0, 0), (1, 0), (0, 0)))[:, 0] * mag0[:, 0]       # :236-239
    mr, mi = r[:, 0], i[:, 0]
    mm = np.sqrt(np.maximum(mr ** 2 + mi ** 2, _UEPS))                     # :244
    rp, ip = mr / (mm + _UEPS), mi / (mm + _UEPS)
    mph = np.arctan2(ip + _UEPS, rp)                                        # :248
    mm = np.pad(np.tanh(mm + _UEPS), ((0, 0), (1, 0), (0, 0)))
    mph = np.pad(mph, ((0, 0), (1, 0), (0, 0)))
    est_m = (mm * mag0[:, 0] + m) * 0.5                                     # :254-262
    est_p = phase + mph
    return est_m * np.cos(est_p), est_m * np.sin(est_p)
