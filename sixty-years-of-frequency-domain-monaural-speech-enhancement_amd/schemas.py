"""State-dict key schemas of the reference models, built programmatically.

A schema is an OrderedDict name -> (shape, 'f32' | 'i64') in the reference's `state_dict()` order; it drives
strict loading and the deterministic synthetic weights (`synth.synth_state_dict`).  tests/test_schemas.py
checks each one against the schema captured from the imported reference (tests/golden/schema_*.json).
"""
from collections import OrderedDict


def _bn(d, p, c):
    d[p + 'weight'] = ((c,), 'f32')
    d[p + 'bias'] = ((c,), 'f32')
    d[p + 'running_mean'] = ((c,), 'f32')
    d[p + 'running_var'] = ((c,), 'f32')
    d[p + 'num_batches_tracked'] = ((), 'i64')


def _lstm(d, p, inp, hid, layers=1, bidir=False, gates=4):
    """torch.nn.LSTM (gates=4) / torch.nn.GRU (gates=3) parameter entries."""
    for k in range(layers):
        i = inp if k == 0 else hid * (2 if bidir else 1)
        for suf in (('', '_reverse') if bidir else ('',)):
            d[f'{p}weight_ih_l{k}{suf}'] = ((gates * hid, i), 'f32')
            d[f'{p}weight_hh_l{k}{suf}'] = ((gates * hid, hid), 'f32')
            d[f'{p}bias_ih_l{k}{suf}'] = ((gates * hid,), 'f32')
            d[f'{p}bias_hh_l{k}{suf}'] = ((gates * hid,), 'f32')


def _conv(d, p, co, ci, k):
    d[p + 'weight'] = ((co, ci) + tuple(k), 'f32')
    d[p + 'bias'] = ((co,), 'f32')


def _deconv(d, p, ci, co, k):
    d[p + 'weight'] = ((ci, co) + tuple(k), 'f32')
    d[p + 'bias'] = ((co,), 'f32')


def lstm_schema():
    """LSTM/LSTM.py:14-22."""
    d = OrderedDict()
    _bn(d, 'bn.', 161)
    _lstm(d, 'lstm1.', 161, 1024, 1)
    _lstm(d, 'lstm2.', 1024, 1024, 2)
    d['fc.0.weight'] = ((161, 1024), 'f32')
    d['fc.0.bias'] = ((161,), 'f32')
    return d


def _crn_like(en_ch, de_ch, lstm_fn, act_prelu):
    d = OrderedDict()
    for i in range(5):
        p = f'en.en_module.{i}.'
        _conv(d, p + '1.', en_ch[i + 1], en_ch[i], (2, 3))
        _bn(d, p + '2.', en_ch[i + 1])
        if act_prelu:
            d[p + '3.weight'] = ((1,), 'f32')
    lstm_fn(d)
    for i in range(5):
        p = f'de.de_module.{i}.'
        _deconv(d, p + '0.', de_ch[i][0], de_ch[i][1], (2, 3))
        off = 3 if i == 3 else 2
        has_bn = de_ch[i][2]
        if has_bn:
            _bn(d, f'{p}{off}.', de_ch[i][1])
            if act_prelu:
                d[f'{p}{off + 1}.weight'] = ((1,), 'f32')
    return d


def crn_schema():
    """CRN/CRN.py:16-109."""
    return _crn_like([1, 16, 32, 64, 128, 256],
                     [(512, 128, True), (256, 64, True), (128, 32, True), (64, 16, True), (32, 1, True)],
                     lambda d: _lstm(d, 'lstm.', 1024, 1024, 2), False)


def dpcrn_schema():
    """DPCRN/DPCRN.py:16-166."""
    def rnn(d):
        _lstm(d, 'dprnn.intra_rnn.', 128, 64, 2, bidir=True)
        d['dprnn.intra_fc.weight'] = ((128, 128), 'f32')
        d['dprnn.intra_fc.bias'] = ((128,), 'f32')
        _lstm(d, 'dprnn.inter_rnn.', 128, 128, 2)
        d['dprnn.inter_fc.weight'] = ((128, 128), 'f32')
        d['dprnn.inter_fc.bias'] = ((128,), 'f32')
        for n in ('ln1', 'ln2'):
            d[f'dprnn.{n}.weight'] = ((4, 128), 'f32')
            d[f'dprnn.{n}.bias'] = ((4, 128), 'f32')
    return _crn_like([2, 32, 32, 32, 64, 128],
                     [(256, 64, True), (128, 32, True), (64, 32, True), (64, 32, True), (64, 2, False)], rnn, True)


def dccrn_schema(kernel_num=(32, 64, 128, 256, 256, 256), rnn_units=256, fft_len=512):
    """DCCRN/DCCRN_cprs.py:47-137 with the decode script's constructor (dccrn_decode_vb.py:11); operator
    sub-keys (`real_conv`, `imag_conv`, `real_lstm`, `imag_lstm`, `r_trans`, `i_trans`) are upstream complexnn's."""
    kn = [2] + list(kernel_num)
    d = OrderedDict()
    for k in range(len(kn) - 1):
        p = f'encoder.{k}.'
        _conv(d, p + '0.real_conv.', kn[k + 1] // 2, kn[k] // 2, (5, 2))
        _conv(d, p + '0.imag_conv.', kn[k + 1] // 2, kn[k] // 2, (5, 2))
        _bn(d, p + '1.', kn[k + 1])
        d[p + '2.weight'] = ((1,), 'f32')
    for n, idx in enumerate(range(len(kn) - 1, 0, -1)):
        p = f'decoder.{n}.'
        _deconv(d, p + '0.real_conv.', kn[idx], kn[idx - 1] // 2, (5, 2))
        _deconv(d, p + '0.imag_conv.', kn[idx], kn[idx - 1] // 2, (5, 2))
        if idx != 1:
            _bn(d, p + '1.', kn[idx - 1])
            d[p + '2.weight'] = ((1,), 'f32')
    # `self.enhance` is registered after both ModuleLists (DCCRN_cprs.py:60-61,94)
    hidden_dim = fft_len // (2 ** len(kn))
    inp = hidden_dim * kn[-1]
    for li in range(2):
        p = f'enhance.{li}.'
        i = (inp if li == 0 else rnn_units) // 2
        _lstm(d, p + 'real_lstm.', i, rnn_units // 2)
        _lstm(d, p + 'imag_lstm.', i, rnn_units // 2)
        if li == 1:
            d[p + 'r_trans.weight'] = ((inp // 2, rnn_units // 2), 'f32')
            d[p + 'r_trans.bias'] = ((inp // 2,), 'f32')
            d[p + 'i_trans.weight'] = ((inp // 2, rnn_units // 2), 'f32')
            d[p + 'i_trans.bias'] = ((inp // 2,), 'f32')
    return d


def fullsubnet_schema(gates=4):
    """FullSubNet/fullsubnet_net_sa/model.py:38-56 with the decode script's sizes (fullsubnet_sa_decode_vb.py:11-24);
    gates=3: `sequence_model="GRU"` (sequence_model.py:36-43)."""
    d = OrderedDict()
    _lstm(d, 'fb_model.sequence_model.', 257, 512, 2, gates=gates)
    d['fb_model.fc_output_layer.weight'] = ((257, 512), 'f32')
    d['fb_model.fc_output_layer.bias'] = ((257,), 'f32')
    _lstm(d, 'sb_model.sequence_model.', 32, 384, 2, gates=gates)
    d['sb_model.fc_output_layer.weight'] = ((2, 384), 'f32')
    d['sb_model.fc_output_layer.bias'] = ((2,), 'f32')
    return d


def fullsubnet_gru_schema():
    return fullsubnet_schema(gates=3)


def gcrn_schema():
    """GCRN/GCRN_noncprs.py:86-133 (registration order of Net.__init__)."""
    d = OrderedDict()
    ec = [2, 16, 32, 64, 128, 256]
    for k in range(1, 6):
        for c in ('conv1', 'conv2'):
            _conv(d, f'conv{k}.{c}.', ec[k], ec[k - 1], (1, 3))
    for lst in ('lstm_list1', 'lstm_list2'):
        for i in range(2):
            _lstm(d, f'glstm.{lst}.{i}.', 512, 512)
    for n in ('ln1', 'ln2'):
        d[f'glstm.{n}.weight'] = ((1024,), 'f32')
        d[f'glstm.{n}.bias'] = ((1024,), 'f32')
    dci, dco = [512, 256, 128, 64, 32], [128, 64, 32, 16, 1]
    for br in (1, 2):
        for i, k in enumerate((5, 4, 3, 2, 1)):
            for c in ('conv1', 'conv2'):
                _deconv(d, f'conv{k}_t_{br}.{c}.', dci[i], dco[i], (1, 3))
    for k in range(1, 6):
        _bn(d, f'bn{k}.', ec[k])
    for br in (1, 2):
        for i, k in enumerate((5, 4, 3, 2, 1)):
            _bn(d, f'bn{k}_t_{br}.', dco[i])
    for br in (1, 2):
        d[f'fc{br}.weight'] = ((161, 161), 'f32')
        d[f'fc{br}.bias'] = ((161,), 'f32')
    return d


def _cts_gate(d, p, idx, co, ci, k, deconv):
    shape = (ci, co) + tuple(k) if deconv else (co, ci) + tuple(k)
    for br in ('conv', 'gate_conv'):
        d[f'{p}0.{br}.{idx}.weight'] = (shape, 'f32')
        d[f'{p}0.{br}.{idx}.bias'] = ((co,), 'f32')
    d[p + '1.weight'] = ((co,), 'f32')
    d[p + '1.bias'] = ((co,), 'f32')
    d[p + '2.weight'] = ((co,), 'f32')


def _cts_glu(d, p, dil, left, right):
    d[p + 'in_conv.weight'] = ((64, 256, 1), 'f32')
    for br in (left, right):
        d[f'{p}{br}.0.weight'] = ((64,), 'f32')
        d[f'{p}{br}.1.weight'] = ((64,), 'f32')
        d[f'{p}{br}.1.bias'] = ((64,), 'f32')
        d[f'{p}{br}.2.weight'] = ((1, 1, 2 * dil - 1), 'f32')
        d[f'{p}{br}.4.weight'] = ((64, 64, 5), 'f32')
    d[p + 'out_conv.0.weight'] = ((64,), 'f32')
    d[p + 'out_conv.1.weight'] = ((64,), 'f32')
    d[p + 'out_conv.1.bias'] = ((64,), 'f32')
    d[p + 'out_conv.2.weight'] = ((256, 64, 1), 'f32')


def cts_step1_schema():
    """CTSNet/Step1_network.py:12-211 `Step1_net()`."""
    d = OrderedDict()
    for i in range(5):
        _cts_gate(d, f'en.en.{i}.', 1, 64, 1 if i == 0 else 64, (2, 5) if i == 0 else (2, 3), False)
    d['de.de6.0.weight'] = ((161, 161), 'f32')       # de6 is registered before the ModuleList (Step1_network.py:109-112)
    d['de.de6.0.bias'] = ((161,), 'f32')
    for i in range(5):
        _cts_gate(d, f'de.de.{i}.', 0, 1 if i == 4 else 64, 128, (2, 5) if i == 4 else (2, 3), True)
    for k in (1, 2, 3):
        for i in range(6):
            _cts_glu(d, f'tcm{k}.tcm_list.{i}.', 2 ** i, 'left_conv', 'right_conv')
    return d


def cts_step2_schema(X=6, R=3):
    """CTSNet/Step2_network.py:13-210 `Step2_net(X=6, R=3)`."""
    d = OrderedDict()
    for i in range(5):
        _cts_gate(d, f'en.en_module.{i}.', 1, 64, 4 if i == 0 else 64, (2, 5) if i == 0 else (2, 3), False)
    for br in ('de_r', 'de_i'):
        for i in range(5):
            _cts_gate(d, f'{br}.de_list.{i}.', 0, 1 if i == 4 else 64, 128, (2, 5) if i == 4 else (2, 3), True)
        d[f'{br}.de6.0.weight'] = ((161, 161), 'f32')
        d[f'{br}.de6.0.bias'] = ((161,), 'f32')
    for r in range(R):
        for i in range(X):
            _cts_glu(d, f'tcm_list.{r}.glu_list.{i}.', 2 ** i, 'ori_conv', 'att_ori')
    return d


def _from_data(name):
    """Large schemas (hundreds of keys) ship as data: the (key, shape, dtype) list captured from the reference
    module's state_dict() by oracle/gen_golden.py - names and shapes only, no weights."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'schema_data', name + '.json')
    with open(path) as f:
        return OrderedDict((k, (tuple(sh), d)) for k, sh, d in json.load(f))


def taylorsenet_schema():
    """TaylorSENet/TaylorSENet.py:8 as built at taylorsenet_decode_vb.py:11-13 (811 keys)."""
    return _from_data('taylorsenet')


def g2net_schema():
    """G2Net_VB/gaf_net_320.py:10 `gaf_base(...)` as built at com_decode.py:23 (stage_num = 3, 825 keys)."""
    return _from_data('g2net')


def uformer_schema():
    """Uformer/uformer.py:30 `Uformer()` (664 keys).  The reference state dict also carries the frozen conv-STFT
    kernels `stft.K, stft.w, istft.K, istft.w` that forward never uses: accepted and ignored."""
    return _from_data('uformer')


def repeat_variant(schema, prefix, n):
    """The schema of a constructor whose ModuleList `prefix<i>.` holds n equal blocks instead of the captured count
    (gaf_base(stage_num = n): `gafs.`, gaf_net_320.py:55-58; TaylorSENet(order_num = n): `highorderblock_list.`,
    TaylorSENet.py:66-70).  Blocks keep their place in state_dict() order; block i >= the captured count copies the last one."""
    blocks, out, emitted = OrderedDict(), OrderedDict(), False
    for k in schema:
        if k.startswith(prefix):
            idx, rest = k[len(prefix):].split('.', 1)
            blocks.setdefault(int(idx), []).append(rest)
    have = len(blocks)
    for k, v in schema.items():
        if not k.startswith(prefix):
            out[k] = v
            continue
        if emitted:
            continue
        emitted = True
        for i in range(n):
            src = min(i, have - 1)
            for rest in blocks[src]:
                out[f'{prefix}{i}.{rest}'] = schema[f'{prefix}{src}.{rest}']
    return out


def cln_variant(schema):
    """The `*_new` flavour of a schema: every InstanceNorm (a 1-D `weight` with a sibling 1-D `bias`) becomes a
    CumulativeLayerNorm with `gain` / `bias` of shape [1,C,1,1] (2-D) or [1,C,1] (inside a TCM), same position
    (CTSNet_new/Step1_network.py:213-286)."""
    out = OrderedDict()
    for k, (shape, dt) in schema.items():
        stem, leaf = k.rsplit('.', 1)
        is_norm = (len(shape) == 1 and len(schema.get(stem + '.weight', ((), ''))[0]) == 1 and stem + '.bias' in schema
                   and len(schema[stem + '.bias'][0]) == 1 and leaf in ('weight', 'bias'))
        if not is_norm:
            out[k] = (shape, dt)
            continue
        one_d = 'tcm' in stem or 'glu_list' in stem
        new_shape = (1, shape[0], 1) if one_d else (1, shape[0], 1, 1)
        out[stem + ('.gain' if leaf == 'weight' else '.bias')] = (new_shape, dt)
    return out


SCHEMAS = {'fullsubnet_gru': fullsubnet_gru_schema, 'taylorsenet': taylorsenet_schema, 'uformer': uformer_schema, 'g2net': g2net_schema, 'cts_step1': cts_step1_schema, 'cts_step2': cts_step2_schema, 'gcrn': gcrn_schema, 'fullsubnet': fullsubnet_schema, 'lstm': lstm_schema, 'crn': crn_schema, 'dpcrn': dpcrn_schema, 'dccrn': dccrn_schema}
SCHEMAS.update({n + '_new': (lambda n=n: cln_variant(SCHEMAS[n]())) for n in ('cts_step1', 'cts_step2', 'taylorsenet', 'g2net')})
