"""Numpy restatement of the torch.nn operators the reference models call.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Semantics follow torch.nn in
eval mode, which is what every reference `*_decode_vb.py` runs
(`model.eval()`, e.g. DCCRN/dccrn_decode_vb.py:14, CRN/crn_decode_vb.py:21).

All functions take/return numpy arrays and compute in the dtype of their
input (tests use float64 as the "truth" and float32 to mirror the reference).
Tensor layouts are the reference's (NCHW, [T,B,F] for batch_first=False LSTM).
"""
import numpy as np


# ----------------------------------------------------------------------------
# activations  (torch.nn.ELU / Softplus / PReLU / Sigmoid / Tanh)
# ----------------------------------------------------------------------------
def sigmoid(x):
    out = np.empty_like(x)
    pos = x >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-x[pos]))
    e = np.exp(x[~pos])
    out[~pos] = e / (1.0 + e)
    return out


def elu(x, alpha=1.0):
    return np.where(x > 0, x, alpha * np.expm1(np.minimum(x, 0)))


def softplus(x, beta=1.0, threshold=20.0):
    # torch.nn.Softplus: linear above threshold
    bx = beta * x
    return np.where(bx > threshold, x, np.log1p(np.exp(np.minimum(bx, threshold))) / beta)


def prelu(x, weight, channel_axis=1):
    """nn.PReLU(): weight shape [1] (shared) or [C] (per channel on axis 1)."""
    w = np.asarray(weight, dtype=x.dtype)
    if w.size > 1:
        shape = [1] * x.ndim
        shape[channel_axis] = w.size
        w = w.reshape(shape)
    return np.where(x >= 0, x, w * x)


def relu(x):
    return np.maximum(x, 0)


# ----------------------------------------------------------------------------
# normalisation (eval mode)
# ----------------------------------------------------------------------------
def batchnorm(x, weight, bias, running_mean, running_var, eps=1e-5, channel_axis=1):
    """nn.BatchNorm{1,2,3}d in eval(): running statistics."""
    shape = [1] * x.ndim
    shape[channel_axis] = -1
    s = (weight / np.sqrt(running_var + eps)).astype(x.dtype)
    return (x - running_mean.reshape(shape).astype(x.dtype)) * s.reshape(shape) + bias.reshape(shape).astype(x.dtype)


def layernorm(x, weight, bias, n_norm_dims, eps=1e-5):
    """nn.LayerNorm over the last n_norm_dims dims (biased variance)."""
    axes = tuple(range(x.ndim - n_norm_dims, x.ndim))
    mu = x.mean(axis=axes, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=axes, keepdims=True)
    y = (x - mu) / np.sqrt(var + eps)
    if weight is not None:
        y = y * weight.astype(x.dtype) + bias.astype(x.dtype)
    return y


def instancenorm(x, weight, bias, eps=1e-5):
    """nn.InstanceNorm{1,2}d(affine=True), track_running_stats=False:
    statistics over all dims after the channel dim, per (b, c), biased var."""
    axes = tuple(range(2, x.ndim))
    mu = x.mean(axis=axes, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=axes, keepdims=True)
    y = (x - mu) / np.sqrt(var + eps)
    if weight is not None:
        shape = [1, -1] + [1] * (x.ndim - 2)
        y = y * weight.reshape(shape).astype(x.dtype) + bias.reshape(shape).astype(x.dtype)
    return y


def cumulative_layernorm(x, gain, bias, eps=1e-5):
    """CumulativeLayerNorm2d / 1d of the `_new` variants (CTSNet_new/Step1_network.py:213-286): at frame t the statistics
    run over all channels (and frequencies) of frames <= t.  x [B,C,T,F] or [B,C,T]; gain / bias [1,C,1(,1)].
    Variance uses the reference's formula (sum x^2 - 2 mu sum x) / n + mu^2."""
    axes = (1, 3) if x.ndim == 4 else (1,)
    per = x.shape[1] * (x.shape[3] if x.ndim == 4 else 1)
    step = x.sum(axis=axes, keepdims=True)
    step2 = (x ** 2).sum(axis=axes, keepdims=True)
    cs, cp = np.cumsum(step, axis=2), np.cumsum(step2, axis=2)
    cnt = (per * np.arange(1, x.shape[2] + 1)).astype(x.dtype)
    cnt = cnt.reshape((1, 1, -1, 1) if x.ndim == 4 else (1, 1, -1))
    mu = cs / cnt
    var = (cp - 2 * mu * cs) / cnt + mu ** 2
    return (x - mu) / np.sqrt(var + eps) * gain.astype(x.dtype) + bias.astype(x.dtype)


# ----------------------------------------------------------------------------
# linear / conv
# ----------------------------------------------------------------------------
def linear(x, weight, bias=None):
    y = x @ weight.T.astype(x.dtype)
    if bias is not None:
        y = y + bias.astype(x.dtype)
    return y


def _pair(v):
    return (v, v) if np.isscalar(v) else tuple(v)


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    """torch.nn.functional.conv2d.  x [B,C,H,W]; weight [Co, C/groups, kh, kw];
    padding = int | (ph, pw) (symmetric zero padding)."""
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw = _pair(dilation)
    B, C, H, W = x.shape
    Co, Cg, kh, kw = weight.shape
    assert C == Cg * groups and Co % groups == 0
    if ph or pw:
        x = np.pad(x, ((0, 0), (0, 0), (ph, ph), (pw, pw)))
    Hp, Wp = x.shape[2], x.shape[3]
    Ho = (Hp - dh * (kh - 1) - 1) // sh + 1
    Wo = (Wp - dw * (kw - 1) - 1) // sw + 1
    out = np.zeros((B, Co, Ho, Wo), dtype=x.dtype)
    w = weight.astype(x.dtype)
    cog = Co // groups
    for g in range(groups):
        xg = x[:, g * Cg:(g + 1) * Cg]
        wg = w[g * cog:(g + 1) * cog]
        acc = out[:, g * cog:(g + 1) * cog]
        for i in range(kh):
            for j in range(kw):
                patch = xg[:, :, i * dh: i * dh + (Ho - 1) * sh + 1: sh,
                           j * dw: j * dw + (Wo - 1) * sw + 1: sw]
                # [Co,Ci] x [B,Ci,Ho,Wo] -> [B,Co,Ho,Wo]
                acc += np.einsum('oc,bchw->bohw', wg[:, :, i, j], patch, optimize=True)
    if bias is not None:
        out += bias.reshape(1, -1, 1, 1).astype(x.dtype)
    return out


def conv_transpose2d(x, weight, bias=None, stride=1, padding=0, output_padding=0):
    """torch.nn.functional.conv_transpose2d (groups=1, dilation=1).
    x [B,Ci,H,W]; weight [Ci, Co, kh, kw]."""
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    oph, opw = _pair(output_padding)
    B, Ci, H, W = x.shape
    Ci2, Co, kh, kw = weight.shape
    assert Ci == Ci2
    Hfull = (H - 1) * sh + kh
    Wfull = (W - 1) * sw + kw
    Ho = Hfull - 2 * ph + oph
    Wo = Wfull - 2 * pw + opw
    full = np.zeros((B, Co, Hfull + oph, Wfull + opw), dtype=x.dtype)
    w = weight.astype(x.dtype)
    for i in range(kh):
        for j in range(kw):
            contrib = np.einsum('co,bchw->bohw', w[:, :, i, j], x, optimize=True)
            full[:, :, i: i + (H - 1) * sh + 1: sh, j: j + (W - 1) * sw + 1: sw] += contrib
    out = full[:, :, ph: ph + Ho, pw: pw + Wo]
    if bias is not None:
        out = out + bias.reshape(1, -1, 1, 1).astype(x.dtype)
    return np.ascontiguousarray(out)


def conv1d(x, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    """x [B,C,L]; weight [Co, C/groups, k]."""
    y = conv2d(x[:, :, None, :], weight[:, :, None, :], bias, (1, stride), (0, padding), (1, dilation), groups)
    return y[:, :, 0, :]


# ----------------------------------------------------------------------------
# LSTM  (torch.nn.LSTM: gate order i,f,g,o; b_ih + b_hh)
# ----------------------------------------------------------------------------
def lstm_layer(x, w_ih, w_hh, b_ih, b_hh, reverse=False):
    """One direction of one nn.LSTM layer, zero initial state.
    x [T, B, I] -> h [T, B, H]."""
    T, B, _ = x.shape
    H = w_hh.shape[1]
    dt = x.dtype
    gx = x @ w_ih.T.astype(dt) + (b_ih + b_hh).astype(dt)          # [T,B,4H]
    whhT = w_hh.T.astype(dt)
    h = np.zeros((B, H), dtype=dt)
    c = np.zeros((B, H), dtype=dt)
    out = np.empty((T, B, H), dtype=dt)
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        g = gx[t] + h @ whhT
        i = sigmoid(g[:, 0:H])
        f = sigmoid(g[:, H:2 * H])
        gg = np.tanh(g[:, 2 * H:3 * H])
        o = sigmoid(g[:, 3 * H:4 * H])
        c = f * c + i * gg
        h = o * np.tanh(c)
        out[t] = h
    return out


def gru_layer(x, w_ih, w_hh, b_ih, b_hh):
    """One torch.nn.GRU layer (gate order r, z, n), zero initial state.  x [T, B, I] -> h [T, B, H]:
    r = s(W_ir x + b_ir + W_hr h + b_hr), z = s(W_iz x + b_iz + W_hz h + b_hz),
    n = tanh(W_in x + b_in + r * (W_hn h + b_hn)), h' = (1 - z) * n + z * h."""
    T, B, _ = x.shape
    H = w_hh.shape[1]
    dt = x.dtype
    gx = x @ w_ih.T.astype(dt) + b_ih.astype(dt)
    whhT, bh = w_hh.T.astype(dt), b_hh.astype(dt)
    h = np.zeros((B, H), dtype=dt)
    out = np.empty((T, B, H), dtype=dt)
    for t in range(T):
        gh = h @ whhT + bh
        r = sigmoid(gx[t][:, 0:H] + gh[:, 0:H])
        z = sigmoid(gx[t][:, H:2 * H] + gh[:, H:2 * H])
        n = np.tanh(gx[t][:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        out[t] = h
    return out


def gru(x, sd, prefix, num_layers=1, batch_first=False):
    """nn.GRU forward (unidirectional) with parameters read from state-dict `sd` under `prefix`."""
    if batch_first:
        x = np.swapaxes(x, 0, 1)
    for k in range(num_layers):
        x = gru_layer(x, sd[f'{prefix}weight_ih_l{k}'], sd[f'{prefix}weight_hh_l{k}'], sd[f'{prefix}bias_ih_l{k}'],
                      sd[f'{prefix}bias_hh_l{k}'])
    if batch_first:
        x = np.swapaxes(x, 0, 1)
    return x


def lstm(x, sd, prefix, num_layers=1, bidirectional=False, batch_first=False):
    """nn.LSTM forward with parameters read from state-dict `sd` under
    `prefix` ('' or 'name.'), key names weight_ih_l{k}[_reverse] etc."""
    if batch_first:
        x = np.swapaxes(x, 0, 1)
    for k in range(num_layers):
        outs = []
        for suf, rev in (('', False), ('_reverse', True)) if bidirectional else (('', False),):
            outs.append(lstm_layer(x,
                                   sd[f'{prefix}weight_ih_l{k}{suf}'], sd[f'{prefix}weight_hh_l{k}{suf}'],
                                   sd[f'{prefix}bias_ih_l{k}{suf}'], sd[f'{prefix}bias_hh_l{k}{suf}'],
                                   reverse=rev))
        x = np.concatenate(outs, axis=-1) if bidirectional else outs[0]
    if batch_first:
        x = np.swapaxes(x, 0, 1)
    return x
