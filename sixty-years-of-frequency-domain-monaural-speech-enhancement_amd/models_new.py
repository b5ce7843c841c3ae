"""Host-side mirrors of the reference's `*_new` directories (CTSNet_new, TaylorSENet_new, G2Net_new).

Those directories re-define the classes of their base directories with every InstanceNorm{1,2}d replaced by a
CumulativeLayerNorm{1,2}d (CTSNet_new/Step1_network.py:213-286: parameters `gain` / `bias` of shape [1,C,1(,1)]) and
decode with the compressed exponents 0.5 / 2.0 (CTSNet_new/two_stage_com_decode_vb.py:73,87,
TaylorSENet_new/taylorsenet_decode_vb.py:40,44, G2Net_new/com_decode.py:53,76).  The class names are the reference's;
the engine picks the norm from the keys of the state dict it is given.
"""
from . import models as _base


class Step1_net(_base.Step1_net):
    """CTSNet_new/Step1_network.py:12."""
    _schema = 'cts_step1_new'
    p_in, p_out = 0.5, 2.0


class Step2_net(_base.Step2_net):
    """CTSNet_new/Step2_network.py:13."""
    _schema = 'cts_step2_new'
    p_in, p_out = 0.5, 2.0


class CTSNet(_base.CTSNet):
    """Both stages of CTSNet_new/two_stage_com_decode_vb.py:13-16 in one engine."""
    _stages = (Step1_net, Step2_net)

    def __init__(self, **kw):
        kw = {k: v for k, v in kw.items() if v is not None}
        kw.setdefault('p_in', 0.5)
        kw.setdefault('p_out', 2.0)
        super().__init__(**kw)


class TaylorSENet(_base.TaylorSENet):
    """TaylorSENet_new/TaylorSENet.py:8."""
    _schema = 'taylorsenet_new'
    p_in, p_out = 0.5, 2.0


class gaf_base(_base.gaf_base):
    """G2Net_new/gaf_net_320.py:10."""
    _schema = 'g2net_new'
    p_in, p_out = 0.5, 2.0


def _taylor(**kw):
    return TaylorSENet(cin=2, k1=(1, 3), k2=(2, 3), c=64, kd1=5, cd1=64, d_feat=256, dilations=[1, 2, 5, 9], p=2,
                       fft_num=320, order_num=3, intra_connect='cat', inter_connect='cat', is_causal=True,
                       is_conformer=False, is_u2=True, is_param_share=False, is_encoder_share=False, **kw)


def _g2net(**kw):
    return gaf_base(3, 64, 2, 4, 4, [1, 2, 5, 9], 256 + 161 * 2, 256, 256, (2, 3), (1, 3), 64, 'cat', 3, is_aux=False,
                    encoder_type='U2Net', tcm_type='full-band', **kw)


_base.MODEL_CLASSES.update({'taylorsenet_new': _taylor, 'g2net_new': _g2net})
