"""SE_ARENA_POISON=1 fills the activation arena with NaN patterns whenever it is re-carved: a kernel that reads a value nobody
wrote (a tail column, a history column of a frame-online window, a padded row) then shows up as NaN instead of depending on what
the previous carve left behind.  Runs in a subprocess (the switch is read once per process)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import se_amd
from se_amd import synth, models_new
from se_amd.models import MODEL_CLASSES
L, B = 12000, 2
x = np.stack([synth.synth_clip(860 + b, 'speech', L) for b in range(B)])
xt = torch.from_numpy(x).cuda()
for name in ('dccrn', 'crn', 'uformer', 'g2net', 'ctsnet_new'):
    if name == 'ctsnet_new':
        m = models_new.CTSNet(max_batch=B, max_samples=L).load_synthetic(17, 18)
    else:
        m = MODEL_CLASSES[name](max_batch=B, max_samples=L).load_synthetic(5)
    ref = m.enhance_batch(xt).cpu().numpy()
    assert np.isfinite(ref).all(), name
    one = m.enhance_batch(xt[:1, :9000].contiguous()).cpu().numpy()           # another carve of the same arena
    assert np.isfinite(one).all(), name
    if name in ('dccrn', 'crn', 'ctsnet_new'):
        eng = m.engine
        for chunk in (1, 16):                                                  # thin and MFMA paths, windows re-carved
            eng.stream_begin(B, c=eng.rms_scale(xt), max_chunk_frames=chunk)
            outs = [eng.stream_push(xt[:, p:p + 4000].contiguous()).cpu().numpy() for p in range(0, L, 4000)]
            outs.append(eng.stream_flush().cpu().numpy())
            got = np.concatenate(outs, axis=1)
            assert np.isfinite(got).all(), (name, chunk)
            e = float(np.sqrt(np.mean((got - ref) ** 2)))
            assert e < 1e-6 + 2e-5 * float(np.sqrt(np.mean(ref ** 2))), (name, chunk, e)
print('POISON-OK')
'''


def test_poisoned_arena_never_reaches_an_output():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SE_ARENA_POISON='1')
    r = subprocess.run([sys.executable, '-c', SCRIPT, root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'POISON-OK' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
