"""CPU: the C++ / OpenMP restatement of the DCCRN decode (oracle/dccrn_cpu.cpp - bench.py's `cpu_baseline`) is pinned by
the SAME reference-generated fixtures as the numpy oracle: tests/golden/dccrn.npz (forward `y`, decodes `enh` / `enh_cprs`
of the imported DCCRN_cprs.py on oracle/_complexnn_recall.py), so the number it times is the reference's arithmetic."""
import numpy as np
import pytest

import se_amd  # noqa: F401
from se_amd import synth
from conftest import load_golden, load_schema, rms


@pytest.fixture(scope='module')
def net():
    from oracle.dccrn_cpu import DccrnCpu
    return DccrnCpu(synth.synth_state_dict(load_schema('dccrn'), 14))


def test_forward_matches_reference_fixture(net):
    G = load_golden('dccrn')
    y = net.forward(G['x'])
    assert y.shape == G['y'].shape and rms(y - G['y']) < 2e-6 * max(rms(G['y']), 1.0), rms(y - G['y'])
    assert np.array_equal(net.forward(G['x'], threads=4), y)          # threads only split independent output rows


@pytest.mark.parametrize('key,p', [('enh', (1.0, 1.0)), ('enh_cprs', (0.5, 2.0))])
def test_decode_matches_reference_fixture(net, key, p):
    G = load_golden('dccrn')
    y = net.enhance(G['wav'], *p)
    assert y.shape == G[key].shape
    e = rms(y - G[key])
    assert e < 1e-4 and e < 2e-5 * max(rms(G[key]), 1e-3), (e, rms(G[key]))


def test_batch_modes_agree_and_match_the_numpy_oracle(net):
    from oracle import decode as D
    x = np.stack([synth.synth_clip(300 + k, 'speech' if k % 2 else 'white', 5003) for k in range(3)])
    a = net.enhance(x, 0.5, 2.0, threads=1, mode=0)
    b = net.enhance(x, 0.5, 2.0, threads=3, mode=1)
    assert np.array_equal(a, b)
    sd = synth.synth_state_dict(load_schema('dccrn'), 14)
    ref = D.enhance_dccrn(sd, x[1], 0.5, 2.0)
    assert a[1].shape == ref.shape and rms(a[1] - ref) < 1e-4 and rms(a[1] - ref) < 2e-5 * max(rms(ref), 1e-3)
