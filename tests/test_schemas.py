"""CPU: programmatic state-dict schemas == schemas captured from the imported reference modules."""
import pytest
import se_amd
from se_amd import schemas
from conftest import load_schema


@pytest.mark.parametrize('name', ['lstm', 'crn', 'dpcrn', 'dccrn', 'fullsubnet', 'fullsubnet_gru', 'gcrn', 'cts_step1', 'cts_step2', 'taylorsenet', 'g2net', 'uformer',
                                  'cts_step1_new', 'cts_step2_new', 'taylorsenet_new', 'g2net_new'])
def test_schema_matches_reference(name):
    ref = load_schema(name)
    mine = schemas.SCHEMAS[name]()
    assert list(mine.keys()) == list(ref.keys())
    for k in ref:
        assert tuple(mine[k][0]) == tuple(ref[k][0]) and mine[k][1] == ref[k][1], k


@pytest.mark.parametrize('name,n', [('g2net_s2', 2), ('g2net_s4', 4), ('g2net_new_s2', 2), ('taylorsenet_o1', 1), ('taylorsenet_o4', 4),
                                    ('taylorsenet_new_o1', 1)])
def test_repeat_count_schema_matches_reference(name, n):
    """gaf_base(stage_num = n) / TaylorSENet(order_num = n): the host classes' key schema for a constructor value the decode
    scripts do not use == the schema captured from the imported reference module built with it."""
    from se_amd import models, models_new
    mod = models_new if '_new' in name else models
    if name.startswith('g2net'):
        m = mod.gaf_base(3, 64, 2, 4, 4, [1, 2, 5, 9], 256 + 161 * 2, 256, 256, (2, 3), (1, 3), 64, 'cat', n, is_aux=False,
                         encoder_type='U2Net', tcm_type='full-band')
    else:
        m = mod.TaylorSENet(cin=2, k1=(1, 3), k2=(2, 3), c=64, kd1=5, cd1=64, d_feat=256, dilations=[1, 2, 5, 9], p=2, fft_num=320,
                            order_num=n, intra_connect='cat', inter_connect='cat', is_causal=True, is_conformer=False, is_u2=True,
                            is_param_share=False, is_encoder_share=False)
    ref, mine = load_schema(name), m.state_dict_schema()
    assert list(mine.keys()) == list(ref.keys())
    for k in ref:
        assert tuple(mine[k][0]) == tuple(ref[k][0]) and mine[k][1] == ref[k][1], k
    assert m._flags & 0xF00 == (n + 1) << 8
    with pytest.raises(NotImplementedError):
        models.gaf_base(stage_num=9, is_aux=False)
    with pytest.raises(NotImplementedError):
        models._taylor.__globals__['TaylorSENet'](kd1=5, inter_connect='cat', order_num=9)


@pytest.mark.parametrize('tag,X,R', [('_x4r2', 4, 2), ('_new_x5r4', 5, 4)])
def test_step2_x_r_schema_matches_reference(tag, X, R):
    """Step2_net(X, R) (CTSNet/Step2_network.py:13-21): key schema and flag bits of a configuration the decode script does not use."""
    from se_amd import models, models_new
    m = (models_new if '_new' in tag else models).Step2_net(X=X, R=R)
    ref, mine = load_schema('cts_step2' + tag), m.state_dict_schema()
    assert list(mine.keys()) == list(ref.keys())
    for k in ref:
        assert tuple(mine[k][0]) == tuple(ref[k][0]) and mine[k][1] == ref[k][1], k
    assert m._flags & 0xFF00 == ((R + 1) << 8) | ((X + 1) << 12)
    assert list(type(m).state_dict_schema().keys()) == list(load_schema('cts_step2' + ('_new' if '_new' in tag else '')).keys())
    for bad in ((7, 3), (6, 9), (0, 3)):
        with pytest.raises(NotImplementedError):
            models.Step2_net(X=bad[0], R=bad[1])
