"""The A/B switches of DESIGN.md 8 select code paths the default decode no longer takes (block-form complex layers, the
interaction of Uformer's branches as its own launch, 128-row pointwise tiles, the round-3 cooperative LSTM kernel, one
cooperative launch per LSTM layer).  They stay in the engine for measurements, so they stay under test: the reference-fixture
suites of the models they touch run once more in a child process with the switches thrown (most are read once per process)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SWITCHES = {
    'SE_DCCRN_GAUSS': '0', 'SE_UF_GAUSS': '0', 'SE_UF_FOLD': '0', 'SE_GC_PW_BM64': '0', 'SE_GC_WIDE128': '0',
    'SE_LSTM_CHUNK': '0', 'SE_COOP16': '0', 'SE_COOP4': '0', 'SE_GC_DBG': '32',
}


@pytest.mark.gpu
def test_reference_fixtures_with_the_ab_switches_thrown():
    env = dict(os.environ, **SWITCHES)
    cmd = [sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider',
           os.path.join(ROOT, 'tests', 'test_gpu_dccrn.py'), os.path.join(ROOT, 'tests', 'test_gpu_uformer.py'),
           os.path.join(ROOT, 'tests', 'test_gpu_models.py')]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
