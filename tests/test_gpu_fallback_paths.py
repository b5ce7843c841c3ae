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


# round 5: the norm folds, the combine epilogue, the matrix-core frequency attention and TaylorSENet's second stream all have
# their round-4 paths behind a switch; the Gauss layers' scratch + combine form is reached with SE_GAUSS_CMB=0 while
# SE_DCCRN_GAUSS / SE_UF_GAUSS stay on (above they are off)
SWITCHES_R5 = {
    'SE_IN_FOLD': '0', 'SE_CLN_STATS': '0', 'SE_CLN_PLANE': '0', 'SE_GAUSS_CMB': '0', 'SE_UF_ATT_F_MFMA': '0', 'SE_TAYLOR_FORK': '0',
    'SE_GCRN_ELU_FOLD': '0',   # GCRN: elu(e_k) as a pass of its own instead of a second store of the encoder layers' epilogue
    'SE_LSTM_SHORT': '0',      # DPCRN's intra-frame BiLSTM as projection GEMM + persistent recurrence (k_lstm.hip) instead of k_lstm_short.hip
}


@pytest.mark.gpu
def test_reference_fixtures_with_the_round5_switches_thrown():
    env = dict(os.environ, **SWITCHES_R5)
    cmd = [sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider',
           os.path.join(ROOT, 'tests', 'test_gpu_full_fixture.py'), os.path.join(ROOT, 'tests', 'test_gpu_new_variants.py'),
           os.path.join(ROOT, 'tests', 'test_gpu_dccrn.py'), os.path.join(ROOT, 'tests', 'test_gpu_uformer.py'),
           os.path.join(ROOT, 'tests', 'test_gpu_b256_fixture.py'), '-k',
           'g2net or taylor or ctsnet or dccrn or uformer or dpcrn or gcrn']
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


# round 6: unit-flattened column tiles of the 64-row layers (GCParams::flat_upr) are the default from 1 024 workgroups on, and
# equal-length batches run with rows of whole 128 B lines (zero-extended: the cLN variants, CRN, GCRN, DPCRN; as ragged rows of one
# length: the InstanceNorm networks) - the batch-256 and 4 s fixtures run all of that; here the same fixtures with plain tiles and
# unpadded rows
SWITCHES_R6 = {'SE_GC_FLAT': '0', 'SE_CLN_PAD': '1', 'SE_IN_PAD': '1', 'SE_G2NET_FORK': '0'}


@pytest.mark.gpu
def test_reference_fixtures_with_the_round6_switches_thrown():
    env = dict(os.environ, **SWITCHES_R6)
    cmd = [sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider',
           os.path.join(ROOT, 'tests', 'test_gpu_b256_fixture.py'), os.path.join(ROOT, 'tests', 'test_gpu_full_fixture.py'),
           '-k', 'g2net or taylor or ctsnet or uformer or dpcrn or gcrn or crn']
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_dccrn_fixtures_with_frames_rounded_to_four_and_to_thirty_two():
    """DCCRN's equal-length batches run on frames rounded to 16 since round 6 (501 -> 512; rounds 3-5: 4): the fixtures - batch 256
    with the reference's clip riding in a row, the full 4 s clip - must not depend on the multiple (the decoder's look-ahead sees
    the same zeros behind every clip's last frame whichever it is)."""
    for mult in ('4', '32'):
        env = dict(os.environ, SE_PAD_FRAMES_TO=mult)
        cmd = [sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider',
               os.path.join(ROOT, 'tests', 'test_gpu_b256_fixture.py'), os.path.join(ROOT, 'tests', 'test_gpu_full_fixture.py'),
               '-k', 'dccrn']
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, mult + ': ' + r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_cln_variants_stream_with_the_round6_push_fusions_off():
    """A one-frame push of the cLN variants carries its residual adds inside the cLN window kernel, exchanges the history of a
    concatenating layer's two sources in one launch and runs its independent chains (TaylorSENet's separate encoder, CTSNet's
    imaginary decoder, the three TCM sequences of a G2Net stage) on auxiliary streams (round 6); the same streamed-vs-offline tests
    with all of that off."""
    env = dict(os.environ, SE_CLN_STREAM_RES='0', SE_STREAM_HIST_PAIR='0', SE_TAYLOR_STREAM_FORK='0', SE_CTSNET_STREAM_FORK='0',
               SE_G2NET_STREAM_FORK='0')
    cmd = [sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider',
           os.path.join(ROOT, 'tests', 'test_gpu_streaming.py'), '-k', 'cln_variants or long_stream']
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


_LSTM_SHORT_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import se_amd
from se_amd.models import dpcrn
B, T = int(sys.argv[1]), int(sys.argv[2])
m = dpcrn(max_batch=B, max_samples=160 * (T - 1) + 320).load_synthetic(13)
rng = np.random.default_rng(5)
x = torch.from_numpy(rng.standard_normal((B, 2, T, 161)).astype(np.float32)).cuda()
y = m(x).cpu().numpy()
np.save(sys.argv[3], y)
"""


@pytest.mark.gpu
@pytest.mark.parametrize('B,T', [(3, 87), (1, 257), (5, 53), (2, 200)])
def test_lstm_short_kernel_against_projection_plus_persistent_path(tmp_path, B, T):
    """`lstm_short_kernel` (DPCRN's intra-frame BiLSTM: 4 steps, both directions, input projection inside) against the
    two-launch form it replaced (`SE_LSTM_SHORT=0`: projection GEMM + `lstm_persist_kernel`) on sequence counts B * T that are
    NOT multiples of its 16-sequence tile (261, 257, 265) and one that is (400), through `se_forward`."""
    import numpy as np
    outs = []
    for k, extra in enumerate(({}, {'SE_LSTM_SHORT': '0'})):
        f = str(tmp_path / ('y%d.npy' % k))
        r = subprocess.run([sys.executable, '-c', _LSTM_SHORT_CHILD % ROOT, str(B), str(T), f], env=dict(os.environ, **extra),
                           cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        outs.append(np.load(f))
    a, b = outs
    assert a.shape == b.shape == (B, 2, T, 161) and np.isfinite(a).all()
    err = float(np.sqrt(np.mean((a - b) ** 2))) / max(float(np.sqrt(np.mean(b ** 2))), 1e-12)
    print('lstm_short vs two-launch form, B', B, 'T', T, 'relative rms', err)
    assert err < 2e-5


@pytest.mark.gpu
def test_taylorsenet_with_the_instancenorm_fold_forced():
    """Round 6: the consumer-side InstanceNorm fold is a per-module decision (unet.h: only where the nested convs have no extent in
    time - G2Net); TaylorSENet's (2, 3) kernels keep the apply pass by default.  SE_IN_FOLD=2 forces the fold - the `NRM` tiles with a
    36-column unit halo only run there - on the 4 s and the batch-256 fixtures."""
    env = dict(os.environ, SE_IN_FOLD='2')
    cmd = [sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider',
           os.path.join(ROOT, 'tests', 'test_gpu_b256_fixture.py'), os.path.join(ROOT, 'tests', 'test_gpu_full_fixture.py'),
           '-k', 'taylor']
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


_SPLIT_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import se_amd
from se_amd import synth
from se_amd.models import MODEL_CLASSES
name, B, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
L = 16000
if name == 'ctsnet':
    from se_amd.models import CTSNet
    m = CTSNet(max_batch=B, max_samples=L).load_synthetic(17, 18)
else:
    m = MODEL_CLASSES[name](max_batch=B, max_samples=L).load_synthetic(1)
x = np.stack([synth.synth_clip(700 + b, 'speech', L) for b in range(B)])
y = m.engine.enhance_batch(torch.from_numpy(x).cuda()).cpu().numpy()
np.save(out, y)
"""


@pytest.mark.gpu
@pytest.mark.parametrize('name,B', [('uformer', 64), ('dpcrn', 65), ('uformer', 67), ('ctsnet', 66), ('taylorsenet', 64), ('dpcrn', 193),
                                    ('uformer', 200)])
def test_two_half_batches_side_by_side_equal_one_batch(tmp_path, name, B):
    """Round 6: Uformer, DPCRN, CTSNet and TaylorSENet decode an equal-length batch of 64 clips or more as two half-batches on two streams (a second
    instance of the model with its own workspace, csrc/engine.hip).  Rows are independent: every row of the split decode - first
    half, second half, odd batch sizes, and the three parts Uformer / DPCRN use from 192 clips on - against the one-batch decode
    (SE_BATCH_SPLIT=0)."""
    import numpy as np
    outs = []
    for split in ('1', '0'):
        out = str(tmp_path / f'{name}_{B}_{split}.npy')
        env = dict(os.environ, SE_BATCH_SPLIT=split)
        r = subprocess.run([sys.executable, '-c', _SPLIT_CHILD % ROOT, name, str(B), out], env=env, cwd=ROOT, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append(np.load(out))
    a, b = outs
    assert a.shape == b.shape and np.isfinite(a).all()
    scale = float(np.sqrt(np.mean(b.astype(np.float64) ** 2)))
    err = float(np.abs(a.astype(np.float64) - b).max())
    # (the two forms run different tile shapes on some layers - the batch enters the tile choice - so not bit-for-bit)
    assert err <= 2e-5 * max(scale, 1e-3), (name, B, err, scale)
