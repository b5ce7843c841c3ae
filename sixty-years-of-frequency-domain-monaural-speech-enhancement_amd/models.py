"""Host-side mirrors of the reference's model classes.

Same class names, constructor arguments, `load_state_dict` key schema, `eval()/cuda()` chaining and
`forward()` tensor shapes as the reference `nn.Module`s, but every forward runs in the HIP engine
(libse_engine.so) - these classes hold no parameters and do no arithmetic themselves.
"""
import functools

import numpy as np

from . import schemas, synth
from .engine import Engine


class _class_or_instance_method:
    """f(cls, self) reachable both as Class.f() (self = None) and instance.f()."""

    def __init__(self, f):
        self.f = f

    def __get__(self, obj, cls):
        return functools.partial(self.f, cls if obj is None else type(obj), obj)


class _EngineModule:
    """Common plumbing: lazy engine creation, strict state-dict load, torch-like call surface."""
    _model = None          # key into _lib.MODEL_IDS
    _schema = None         # key into schemas.SCHEMAS (defaults to _model)
    _repeat_prefix = None  # ModuleList whose length is a constructor argument (stage_num / order_num), decode-script value 3
    p_in = 1.0             # magnitude exponents the decode script applies around the network
    p_out = 1.0

    def __init__(self, device=0, max_batch=1, max_samples=64000, p_in=None, p_out=None, graphs=False, flags=0):
        self._flags = flags            # model-specific SE_CFG_* bits (include/se_engine.h)
        self._graphs = graphs          # replay enhance_batch as a hipGraph per shape (small, launch-bound batches)
        self._device = device
        self._max_batch = max_batch
        self._max_samples = max_samples
        if p_in is not None:
            self.p_in = p_in
        if p_out is not None:
            self.p_out = p_out
        self.engine = None

    # -- reference-compatible surface -------------------------------------------------------------
    @_class_or_instance_method
    def state_dict_schema(cls, self):
        """Class.state_dict_schema(): the decode script's configuration; instance.state_dict_schema(): this instance's."""
        base = schemas.SCHEMAS[cls._schema or cls._model]()
        n = getattr(self, '_repeats', 3)
        return base if cls._repeat_prefix is None or n == 3 else schemas.repeat_variant(base, cls._repeat_prefix, n)

    def load_state_dict(self, sd, strict=True):
        assert strict, "the engine only supports strict loads (as every reference decode script does)"
        want = self.state_dict_schema()
        missing = [k for k in want if k not in sd]
        unexpected = [k for k in sd if k not in want]
        if missing or unexpected:
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:5]} unexpected {unexpected[:5]}")
        self.engine = Engine(self._model, self._device, self._max_batch, self._max_samples, self.p_in, self.p_out,
                             graphs=self._graphs, flags=self._flags)
        self.engine.load_state_dict(sd)
        return self

    def load_synthetic(self, seed=0):
        """Deterministic random-init weights (no checkpoints ship with the reference)."""
        return self.load_state_dict(synth.synth_state_dict(self.state_dict_schema(), seed))

    def eval(self):
        return self

    def cuda(self, device=None):
        return self

    def to(self, *a, **k):
        return self

    def __call__(self, x):
        return self.forward(x)

    def forward(self, x):
        if self.engine is None:
            raise RuntimeError("load_state_dict() must be called before forward()")
        return self.engine.forward(x.contiguous())

    def enhance_batch(self, wav):
        return self.engine.enhance_batch(wav)

    def enhance_ragged(self, wav, lengths):
        return self.engine.enhance_ragged(wav, lengths)


class DCCRN(_EngineModule):
    """DCCRN/DCCRN_cprs.py:8 as built by DCCRN/dccrn_decode_vb.py:11.  forward: [B,2,257,T] -> [B,2,257,T]."""
    _model = 'dccrn'

    def __init__(self, rnn_layers=2, rnn_units=128, win_len=512, win_inc=128, fft_len=512, win_type='hanning',
                 masking_mode='E', use_clstm=False, use_cbn=False, kernel_size=5,
                 kernel_num=(16, 32, 64, 128, 256, 256), **kw):
        cfg = (rnn_layers, rnn_units, win_len, win_inc, fft_len, masking_mode, use_clstm, use_cbn, kernel_size,
               tuple(kernel_num))
        if cfg[:5] + cfg[6:] != (2, 256, 512, 128, 512, True, False, 5, (32, 64, 128, 256, 256, 256)) or masking_mode not in ('E', 'C', 'R'):
            raise NotImplementedError("the engine builds the decode script's DCCRN configuration "
                                      "(dccrn_decode_vb.py:11) with masking_mode 'E', 'C' or 'R'; got " + repr(cfg))
        # masking_mode (DCCRN_cprs.py:205-223): SE_CFG_DCCRN_MASK_C = 32, SE_CFG_DCCRN_MASK_R = 64 (include/se_engine.h)
        kw['flags'] = kw.get('flags', 0) | {'E': 0, 'C': 32, 'R': 64}[masking_mode]
        super().__init__(**kw)


class lstm_net(_EngineModule):
    """LSTM/LSTM.py:14 `lstm_net()`.  forward: magnitude [B,T,161] -> enhanced magnitude [B,T,161]."""
    _model = 'lstm'


class crn_net(_EngineModule):
    """CRN/CRN.py:16 `crn_net()`.  forward: magnitude [B,T,161] -> enhanced magnitude [B,T,161]."""
    _model = 'crn'


class dpcrn(_EngineModule):
    """DPCRN/DPCRN.py:16 `dpcrn()`.  forward: RI [B,2,T,161] -> masked RI [B,2,T,161]."""
    _model = 'dpcrn'


class Model(_EngineModule):
    """FullSubNet/fullsubnet_net_sa/model.py:9 `Model(...)` as built at fullsubnet_sa_decode_vb.py:11-24.
    forward: magnitude [B,1,257,T] -> complex mask [B,2,257,T]; each utterance gets batch-1 semantics.
    sequence_model: "LSTM" (the decode script's) or "GRU" (sequence_model.py:36-43; SE_CFG_FSN_GRU); norm_type:
    "offline_laplace_norm" (the decode script's) or "cumulative_laplace_norm" (base_model.py:212-240; SE_CFG_FSN_CUMULATIVE)."""
    _model = 'fullsubnet'
    SE_CFG_FSN_GRU = 8
    SE_CFG_FSN_CUMULATIVE = 16

    def __init__(self, num_freqs=257, look_ahead=2, sequence_model="LSTM", fb_num_neighbors=0, sb_num_neighbors=15,
                 fb_output_activate_function="ReLU", sb_output_activate_function=None, fb_model_hidden_size=512,
                 sb_model_hidden_size=384, norm_type="offline_laplace_norm", num_groups_in_drop_band=2,
                 weight_init=True, **kw):
        cfg = (num_freqs, look_ahead, fb_num_neighbors, sb_num_neighbors, fb_output_activate_function,
               sb_output_activate_function, fb_model_hidden_size, sb_model_hidden_size, norm_type)
        if (cfg[:-1] != (257, 2, 0, 15, "ReLU", None, 512, 384) or sequence_model not in ("LSTM", "GRU")
                or norm_type not in ("offline_laplace_norm", "cumulative_laplace_norm")):
            raise NotImplementedError("the engine builds the decode script's FullSubNet configuration (sequence model LSTM "
                                      "or GRU, norm offline_laplace_norm or cumulative_laplace_norm); got "
                                      + repr(cfg + (sequence_model,)))
        if norm_type == "cumulative_laplace_norm":      # base_model.py:212-240: causal - the engine then also streams (se_stream_*)
            kw['flags'] = kw.get('flags', 0) | self.SE_CFG_FSN_CUMULATIVE
        if sequence_model == "GRU":
            kw['flags'] = kw.get('flags', 0) | self.SE_CFG_FSN_GRU
            self.__class__ = _ModelGRU            # same engine model, the GRU's [3H, .] key schema
        super().__init__(**kw)

    def forward(self, x):
        B, _, F, T = x.shape
        return self.engine.forward(x.contiguous(), out_shape=(B, 2, F, T))


class _ModelGRU(Model):
    _schema = 'fullsubnet_gru'


class Net(_EngineModule):
    """GCRN/GCRN_noncprs.py:86 `Net()`.  forward: RI [B,2,T,161] -> RI estimate [B,2,T,161] (complex mapping).
    The decode script is checked in with the compressed exponents (gcrn_decode_vb.py:40,51)."""
    _model = 'gcrn'
    p_in = 0.5
    p_out = 2.0


class _CtsStage(_EngineModule):
    """One CTSNet stage; the engine model `ctsnet` holds both stages under the key prefixes step1. / step2."""
    _model = 'ctsnet'
    _prefix = ''

    def load_state_dict(self, sd, strict=True):
        want = self.state_dict_schema()
        missing = [k for k in want if k not in sd]
        unexpected = [k for k in sd if k not in want]
        if missing or unexpected:
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:5]} unexpected {unexpected[:5]}")
        self._sd = {self._prefix + k: v for k, v in sd.items()}
        self.engine = Engine('ctsnet', self._device, self._max_batch, self._max_samples, self.p_in, self.p_out,
                             graphs=self._graphs, flags=self._flags)
        self.engine.load_state_dict(self._sd)
        return self


class Step1_net(_CtsStage):
    """CTSNet/Step1_network.py:12 `Step1_net()`.  forward: magnitude [B,T,161] -> magnitude [B,T,161]."""
    _prefix = 'step1.'
    _schema = 'cts_step1'


class Step2_net(_CtsStage):
    """CTSNet/Step2_network.py:13 `Step2_net(X=6, R=3)`.  forward: [B,4,T,161] -> [B,2,T,161]."""
    _prefix = 'step2.'
    _schema = 'cts_step2'

    def __init__(self, X=6, R=3, **kw):
        if not (1 <= X <= 6 and 1 <= R <= 8):
            raise NotImplementedError("the engine builds Step2_net(X, R) with X in [1, 6] and R in [1, 8] (the decode "
                                      "script's: X=6, R=3)")
        # R groups of X gated blocks (Step2_network.py:13-21): SE_CFG_REPEATS(R) | SE_CFG_REPEATS2(X) (include/se_engine.h)
        self._xr = (X, R)
        if (X, R) != (6, 3):
            kw['flags'] = kw.get('flags', 0) | ((R + 1) << 8) | ((X + 1) << 12)
        super().__init__(**kw)

    @_class_or_instance_method
    def state_dict_schema(cls, self):
        X, R = getattr(self, '_xr', (6, 3))
        base = schemas.cts_step2_schema(X, R)
        return schemas.cln_variant(base) if (cls._schema or '').endswith('_new') else base

    def forward(self, x):
        B, _, T, F = x.shape
        return self.engine.forward(x.contiguous(), out_shape=(B, 2, T, F))


class CTSNet:
    """The two chained stages of CTSNet/two_stage_com_decode_vb.py:13-16,78-84 in one engine (decode path)."""

    _stages = (Step1_net, Step2_net)

    def __init__(self, X=6, R=3, **kw):
        self._kw = {k: v for k, v in kw.items() if v is not None}
        self._stage2 = self._stages[1](X=X, R=R)      # Step2_net(X, R): validates, knows its key schema and flag bits
        self.engine = None

    def load_state_dicts(self, sd1, sd2):
        kw = dict(self._kw)
        sd = {'step1.' + k: v for k, v in sd1.items()}
        sd.update({'step2.' + k: v for k, v in sd2.items()})
        self.engine = Engine('ctsnet', kw.get('device', 0), kw.get('max_batch', 1), kw.get('max_samples', 64000),
                             kw.get('p_in', 1.0), kw.get('p_out', 1.0), graphs=kw.get('graphs', False),
                             flags=kw.get('flags', 0) | self._stage2._flags)
        self.engine.load_state_dict(sd)
        return self

    def load_synthetic(self, seed1=17, seed2=18):
        return self.load_state_dicts(synth.synth_state_dict(self._stages[0].state_dict_schema(), seed1),
                                     synth.synth_state_dict(self._stage2.state_dict_schema(), seed2))

    def enhance_batch(self, wav):
        return self.engine.enhance_batch(wav)

    def enhance_ragged(self, wav, lengths):
        return self.engine.enhance_ragged(wav, lengths)


class TaylorSENet(_EngineModule):
    """TaylorSENet/TaylorSENet.py:8 as built at taylorsenet_decode_vb.py:11-13.  forward: RI [B,2,T,161] -> RI."""
    _model = 'taylorsenet'
    _repeat_prefix = 'highorderblock_list.'

    def __init__(self, cin=2, k1=(1, 3), k2=(2, 3), c=64, kd1=3, cd1=64, d_feat=256, dilations=(1, 2, 5, 9), p=2,
                 fft_num=320, order_num=3, intra_connect='cat', inter_connect='add', is_causal=True, is_conformer=False,
                 is_u2=True, is_param_share=False, is_encoder_share=False, **kw):
        cfg = (cin, tuple(k1), tuple(k2), c, kd1, cd1, d_feat, tuple(dilations), p, fft_num, order_num, intra_connect,
               inter_connect, is_causal, is_conformer, is_u2, is_param_share, is_encoder_share)
        if (cfg[:10] + cfg[11:] != (2, (1, 3), (2, 3), 64, 5, 64, 256, (1, 2, 5, 9), 2, 320, 'cat', 'cat', True, False, True, False,
                                     False) or not 0 <= order_num <= 8):
            raise NotImplementedError("the engine builds the decode script's TaylorSENet configuration with order_num in "
                                      "[0, 8]; got " + repr(cfg))
        # order_num (TaylorSENet.py:27,66-70): SE_CFG_REPEATS(n) = (n + 1) << 8 (include/se_engine.h)
        self._repeats = order_num
        if order_num != 3:
            kw['flags'] = kw.get('flags', 0) | ((order_num + 1) << 8)
        super().__init__(**kw)



def _taylor(**kw):
    return TaylorSENet(cin=2, k1=(1, 3), k2=(2, 3), c=64, kd1=5, cd1=64, d_feat=256, dilations=[1, 2, 5, 9], p=2,
                       fft_num=320, order_num=3, intra_connect='cat', inter_connect='cat', is_causal=True,
                       is_conformer=False, is_u2=True, is_param_share=False, is_encoder_share=False, **kw)


class gaf_base(_EngineModule):
    """G2Net_VB/gaf_net_320.py:10 as built at com_decode.py:23.  forward: RI [B,2,T,161] -> list of stage outputs;
    the engine returns the list with only the LAST stage ([B,2,161,T]) materialised - the decode script uses [-1]."""
    _model = 'g2net'
    _repeat_prefix = 'gafs.'

    def __init__(self, kd1=3, cd1=64, tcm_num=2, sub_g1=4, sub_g2=4, dilas=(1, 2, 5, 9), ci=256 + 161 * 2, co1=256,
                 co2=256, k1=(2, 3), k2=(1, 3), c=64, intra_connect='cat', stage_num=3, is_causal=True, is_aux=True,
                 encoder_type='U2Net', tcm_type='full-band', **kw):
        cfg = (kd1, cd1, tcm_num, tuple(dilas), ci, co1, co2, tuple(k1), tuple(k2), c, intra_connect, stage_num,
               is_causal, is_aux, encoder_type, tcm_type)
        if (cfg[:11] + cfg[12:] != (3, 64, 2, (1, 2, 5, 9), 578, 256, 256, (2, 3), (1, 3), 64, 'cat', True, False, 'U2Net', 'full-band')
                or not 1 <= stage_num <= 8):
            raise NotImplementedError("the engine builds the decode script's gaf_base configuration with stage_num in [1, 8]; "
                                      "got " + repr(cfg))
        # stage_num (gaf_net_320.py:27,55-58): SE_CFG_REPEATS(n) = (n + 1) << 8 (include/se_engine.h)
        self._repeats = stage_num
        if stage_num != 3:
            kw['flags'] = kw.get('flags', 0) | ((stage_num + 1) << 8)
        super().__init__(**kw)


    def forward(self, x):
        B, _, T, F = x.shape
        return [self.engine.forward(x.contiguous(), out_shape=(B, 2, F, T))]


def _g2net(**kw):
    return gaf_base(3, 64, 2, 4, 4, [1, 2, 5, 9], 256 + 161 * 2, 256, 256, (2, 3), (1, 3), 64, 'cat', 3, is_aux=False,
                    encoder_type='U2Net', tcm_type='full-band', **kw)


class Uformer(_EngineModule):
    """Uformer/uformer.py:30 `Uformer()`.  forward(inputs, src): waveforms [B, L] -> the reference's 4-tuple
    (enhanced waveform [B, 160*floor(L/160)], istft(stft(src)), output_cplx [B,2,257,T], src_cplx [B,2,257,T]) - the STFT /
    iSTFT are inside the model (uformer.py:178, 276).  Without `src` the two source outputs are None (the decode script
    only reads [0], uformer_decode_vb.py:40)."""
    _model = 'uformer'
    _IGNORED = ('stft.K', 'stft.w', 'istft.K', 'istft.w')

    def __init__(self, win_len=400, win_inc=160, fft_len=512, win_type='hanning', fid=None, **kw):
        if (win_len, win_inc, fft_len) != (400, 160, 512):
            raise NotImplementedError("the engine builds Uformer() with its default front end (400/160/512)")
        super().__init__(**kw)

    def load_state_dict(self, sd, strict=True):
        return super().load_state_dict({k: v for k, v in sd.items() if k not in self._IGNORED}, strict)

    def forward(self, inputs, src=None, spectra=True):
        if self.engine is None:
            raise RuntimeError("load_state_dict() must be called before forward()")
        return self.engine.uformer_forward(inputs.contiguous(), None if src is None else src.contiguous(), spectra)

    def __call__(self, inputs, src=None, spectra=True):
        return self.forward(inputs, src, spectra)


MODEL_CLASSES = {'fullsubnet': Model, 'uformer': Uformer, 'g2net': _g2net, 'taylorsenet': _taylor, 'gcrn': Net, 'lstm': lstm_net, 'crn': crn_net, 'dpcrn': dpcrn,
                 'dccrn': lambda **kw: DCCRN(rnn_units=256, masking_mode='E', use_clstm=True,
                                             kernel_num=[32, 64, 128, 256, 256, 256], **kw)}
