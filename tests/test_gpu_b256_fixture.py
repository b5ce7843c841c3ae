"""GPU: every network at the batch its throughput figures are quoted on (256 clips of 4 s; FullSubNet 128 - its
257 * B sub-band sequences at 256 do not fit the arena) with the clip of the reference-generated full-size fixture riding
in one row, asserted at 1e-4 RMS against the REFERENCE's own decode of that clip (tests/golden/full_<name>.npz,
dccrn.npz:enh4_cprs, dpcrn.npz:enh_real_cprs - oracle/gen_golden.py, imported reference).

Why: `gc_launch` picks its tile from the grid size (64 x 256 above 6 144 workgroups, two-row tiles above 4 096, re-chunked
64-row layers, 128 x 32 tails), `blocks.h` switches a TCM block to `tcm_fused_kernel` from batch 96, the cooperative LSTM
walks several sequence tiles per workgroup above batch 64 - kernel variants a batch of 3 never launches.  Here each of
them runs on the layer shapes, chunkings and epilogues (GLU, residual, InstanceNorm statistics) of every model at T = 401
/ 501 / 251 and is compared with the reference, not with another tiling of the same engine.  Two more rows are compared
with a batch-of-2 engine (pinned by tests/test_gpu_full_fixture.py) so that a row-dependent fault cannot hide in the rows
the fixture does not ride in."""
import numpy as np
import pytest

import se_amd  # noqa: F401
from se_amd import synth
from conftest import load_golden, rms

pytestmark = pytest.mark.gpu
L = 64000
SEEDS = {'lstm': 11, 'crn': 12, 'fullsubnet': 15, 'gcrn': 16, 'taylorsenet': 19, 'g2net': 20, 'uformer': 21,
         'taylorsenet_new': 19, 'g2net_new': 20, 'ctsnet': (17, 18), 'ctsnet_new': (17, 18), 'dccrn': 14, 'dpcrn': None}
BATCH = {'fullsubnet': 128}


def make(name, max_batch, max_samples=L):
    from se_amd import models, models_new  # noqa: F401
    kw = dict(max_batch=max_batch, max_samples=max_samples, p_in=0.5, p_out=2.0)
    if name == 'uformer':
        kw = dict(max_batch=max_batch, max_samples=max_samples)      # in-model STFT; no exponents in its script
    if name.startswith('ctsnet'):
        cls = models_new.CTSNet if name.endswith('_new') else models.CTSNet
        return cls(**kw).load_synthetic(*SEEDS[name])
    m = models.MODEL_CLASSES[name](**kw)
    if name == 'dpcrn':                                              # the reference's REAL compressed-spectrum checkpoint
        return m.load_state_dict(dict(load_golden('ckpt_vb_dpcrn_cprs')))
    return m.load_synthetic(SEEDS[name])


def fixture_clip(name):
    """(clip, reference output) of the 4 s fixture of `name`."""
    if name == 'dccrn':
        return synth.synth_clip(1, 'speech', L), load_golden('dccrn')['enh4_cprs']
    if name == 'dpcrn':
        return synth.synth_clip(0, 'speech', L), load_golden('dpcrn')['enh_real_cprs']
    G = load_golden('full_' + name)
    assert int(G['n']) == L
    return synth.synth_clip(int(G['seed']), 'speech', L), G['enh4_cprs']


@pytest.mark.parametrize('name', sorted(SEEDS))
def test_fixture_row_at_sweep_batch(name):
    import torch
    B = BATCH.get(name, 256)
    base = synth.synth_batch(16, 'speech', L, seed0=700)
    x = np.tile(base, ((B + 15) // 16, 1))[:B].copy()
    x[3::16] = synth.synth_clip(77, 'white', L)
    x[1::16] *= 0.37
    row = 5 + 16 * ((B // 16) // 2)                     # a row in the middle of the batch
    clip, ref = fixture_clip(name)
    x[row] = clip
    big = make(name, B)
    xt = torch.from_numpy(x).cuda()
    y = big.enhance_batch(xt)
    assert bool(torch.isfinite(y).all())
    got = y[row].cpu().numpy()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    e = rms(got - ref)
    print(name, 'B', B, 'fixture row', row, 'rms err vs reference', e, 'rms ref', rms(ref))
    assert e < 1e-4 and e < 5e-4 * max(rms(ref), 1e-3), (name, e, rms(ref))
    small = make(name, 2)
    for k in (0, B - 2):
        ys = small.enhance_batch(xt[k:k + 2])
        for j in (0, 1):
            r = ys[j].cpu().numpy()
            assert rms(y[k + j].cpu().numpy() - r) < 2e-5 * max(rms(r), 1e-4), (name, k + j)
