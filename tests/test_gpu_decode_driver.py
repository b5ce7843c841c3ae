"""GPU: the file-level `enhance(args)` driver (se_amd/decode.py) against the numpy oracle - directory in, PCM_16 WAV
files out, same file names, both flavours of the reference drivers (`*_decode_vb.py` and the WSJ grid `*_decode.py`)."""
import os
import types

import numpy as np
import pytest

import se_amd
from se_amd import synth, schemas, wavio, decode
from conftest import rms

pytestmark = pytest.mark.gpu


def _write_clips(d, lengths, seed0):
    os.makedirs(d, exist_ok=True)
    clips = {}
    for i, L in enumerate(lengths):
        x = synth.synth_clip(seed0 + i, 'speech', L)
        name = f'p{232 + i}_{i:03d}.wav'
        wavio.write_wav_pcm16(os.path.join(d, name), x, 16000)
        clips[name] = wavio.read_wav(os.path.join(d, name))[0]        # what the driver will read back (PCM_16 rounded)
    return clips


def _pcm16(y):
    return np.clip(np.rint(np.asarray(y, dtype=np.float64) * 32767.0), -32768, 32767).astype(np.int64)       # wavio.pcm16_bytes


def test_vb_driver_matches_oracle_and_pcm16(tmp_path):
    import torch
    assert torch.cuda.is_available()
    from oracle import decode as D
    mix, out = str(tmp_path / 'noisy'), str(tmp_path / 'enh')
    clips = _write_clips(mix, [4000, 6000, 4000], 40)
    sd = synth.synth_state_dict(schemas.crn_schema(), 12)
    args = types.SimpleNamespace(mix_file_path=mix, esti_clean_file_path=out, fs=16000)
    n = decode.enhance(args, 'crn', state_dict=sd, max_batch=2)
    assert n == 3 and sorted(os.listdir(out)) == sorted(clips)
    for name, x in clips.items():
        y, fs = wavio.read_wav(os.path.join(out, name))
        ref = D.enhance_crn(sd, x.astype(np.float64))
        assert fs == 16000 and len(y) == len(ref)
        # the files are PCM_16 (soundfile's default subtype): the engine and the reference path must quantise to the
        # same integers up to ties at the rounding boundary -> identical PESQ / STOI by construction
        diff = np.abs(_pcm16(ref) - np.round(y * 32768.0).astype(np.int64))
        assert diff.max() <= 1 and (diff != 0).mean() < 1e-3, (name, diff.max(), (diff != 0).mean())
        assert rms(y - ref) < 1e-4


def test_wsj_grid_driver_paths(tmp_path):
    """CRN/crn_decode.py:28-32: <mix>/<noise_type>/<seen>/<snr>/ in, same sub-tree out."""
    mix, out = tmp_path / 'mix', tmp_path / 'esti'
    cell = os.path.join('cafe', 'unseen', '-5')
    clips = _write_clips(str(mix / cell), [3200, 3200], 50)
    sd = synth.synth_state_dict(schemas.lstm_schema(), 11)
    args = types.SimpleNamespace(mix_file_path=str(mix), esti_clean_file_path=str(out), fs=16000, noise_type='cafe',
                                 seen='unseen', snr='-5')
    assert decode.enhance(args, 'lstm', state_dict=sd, max_batch=4) == 2
    assert sorted(os.listdir(str(out / cell))) == sorted(clips)


def test_ctsnet_driver_two_state_dicts(tmp_path):
    from oracle import decode as D
    mix, out = str(tmp_path / 'noisy'), str(tmp_path / 'enh')
    clips = _write_clips(mix, [4000], 60)
    sd1 = synth.synth_state_dict(schemas.SCHEMAS['cts_step1_new'](), 17)
    sd2 = synth.synth_state_dict(schemas.SCHEMAS['cts_step2_new'](), 18)
    args = types.SimpleNamespace(mix_file_path=mix, esti_file_path=out, fs=16000)
    assert decode.enhance(args, 'ctsnet_new', state_dict=(sd1, sd2), max_batch=1) == 1     # exponents default to 0.5 / 2.0
    (name, x), = clips.items()
    y, _ = wavio.read_wav(os.path.join(out, name))
    ref = D.enhance_ctsnet(sd1, sd2, x.astype(np.float64), 0.5, 2.0)
    assert rms(y - ref) < 1e-4 and np.abs(_pcm16(ref) - np.round(y * 32768.0).astype(np.int64)).max() <= 1


def test_vb_driver_batches_clips_of_different_lengths(tmp_path):
    """A VoiceBank+DEMAND-like directory: every clip has its own length.  The driver must decode them in a few ragged
    calls (not one call per clip) and every file must equal the oracle's per-clip decode.  CTSNet: the model whose
    InstanceNorm makes zero-padding non-neutral."""
    from oracle import decode as D
    mix, out = str(tmp_path / 'noisy'), str(tmp_path / 'enh')
    lengths = [5000, 3210, 4444, 6100, 3999, 5001, 4800]
    clips = _write_clips(mix, lengths, 70)
    sd1 = synth.synth_state_dict(schemas.SCHEMAS['cts_step1'](), 17)
    sd2 = synth.synth_state_dict(schemas.SCHEMAS['cts_step2'](), 18)
    args = types.SimpleNamespace(mix_file_path=mix, esti_file_path=out, fs=16000)
    calls = []
    from se_amd.engine import Engine
    orig_r, orig_b = Engine.enhance_ragged, Engine.enhance_batch
    Engine.enhance_ragged = lambda self, wav, lens, out=None: (calls.append(len(lens)), orig_r(self, wav, lens, out))[1]
    Engine.enhance_batch = lambda self, wav, out=None: (calls.append(wav.shape[0]), orig_b(self, wav, out))[1]
    try:
        assert decode.enhance(args, 'ctsnet', state_dict=(sd1, sd2), max_batch=4, p_in=0.5, p_out=2.0) == len(lengths)
    finally:
        Engine.enhance_ragged, Engine.enhance_batch = orig_r, orig_b
    assert sorted(calls) == [3, 4], calls                       # 7 distinct lengths in two engine calls
    for name, x in clips.items():
        y, _ = wavio.read_wav(os.path.join(out, name))
        ref = D.enhance_ctsnet(sd1, sd2, x.astype(np.float64), 0.5, 2.0)
        assert len(y) == len(ref) and rms(y - ref) < 1e-4, (name, rms(y - ref))
        assert np.abs(_pcm16(ref) - np.round(y * 32768.0).astype(np.int64)).max() <= 1


def _crn_dir(tmp_path, lengths, seed0=80):
    mix = str(tmp_path / 'noisy')
    clips = _write_clips(mix, lengths, seed0)
    sd = synth.synth_state_dict(schemas.crn_schema(), 12)
    return mix, clips, sd


def test_two_ranks_write_disjoint_complete_outputs(tmp_path):
    """One process per GPU: every rank derives its share of the clip list from (rank, world) alone, reads / decodes / writes
    only those clips, and the union is the whole directory, byte for byte what one rank writes."""
    lengths = [4000, 5200, 3300, 6100, 4800, 3900, 5600, 4100, 3000, 5000, 4444]
    mix, clips, sd = _crn_dir(tmp_path, lengths)
    one = str(tmp_path / 'one')
    args = types.SimpleNamespace(mix_file_path=mix, esti_clean_file_path=one, fs=16000)
    assert decode.enhance(args, 'crn', state_dict=sd, max_batch=4, verbose=False) == len(lengths)
    outs = []
    for r in range(2):
        d = str(tmp_path / f'rank{r}')
        args = types.SimpleNamespace(mix_file_path=mix, esti_clean_file_path=d, fs=16000)
        st = {}
        n = decode.enhance(args, 'crn', state_dict=sd, max_batch=4, verbose=False, rank=r, world=2, stats=st)
        outs.append(set(os.listdir(d)))
        assert n == len(outs[-1]) == st['files_rank'] and st['world'] == 2
        for f in outs[-1]:
            assert open(os.path.join(d, f), 'rb').read() == open(os.path.join(one, f), 'rb').read(), f
    assert not (outs[0] & outs[1]) and (outs[0] | outs[1]) == set(clips) and abs(len(outs[0]) - len(outs[1])) <= 1


def test_two_launched_ranks_share_one_output_directory(tmp_path):
    """The command-line form under the launcher (tools/decode_vb.py, RANK / WORLD_SIZE from torch.distributed.run; both ranks
    share the box's one GPU here): the output directory ends up complete."""
    import subprocess
    import sys
    lengths = [4000, 5200, 3300, 6100, 4800, 3900, 5600]
    mix, clips, sd = _crn_dir(tmp_path, lengths, 90)
    ck = str(tmp_path / 'crn.npz')
    np.savez(ck, **sd)
    out = str(tmp_path / 'enh')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                        '127.0.0.1', '--master-port', '29631', os.path.join(root, 'tools', 'decode_vb.py'), '--model', 'crn',
                        '--mix_file_path', mix, '--esti_clean_file_path', out, '--Model_path', ck, '--max_batch', '3'],
                       env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert sorted(os.listdir(out)) == sorted(clips)
    from oracle import decode as D
    name = sorted(clips)[3]
    y, _ = wavio.read_wav(os.path.join(out, name))
    assert rms(y - D.enhance_crn(sd, clips[name].astype(np.float64))) < 1e-4


def test_48k_corpus_through_the_pipeline(tmp_path):
    """VoiceBank+DEMAND ships at 48 kHz: raw PCM_16 -> device -> float -> 48 -> 16 kHz (se_resample) -> ragged decode ->
    device-side PCM_16, against the oracle's resampler + decode of what the files hold."""
    from oracle import decode as D
    from oracle import resample as R
    mix, out = str(tmp_path / 'noisy'), str(tmp_path / 'enh')
    os.makedirs(mix)
    sd = synth.synth_state_dict(schemas.crn_schema(), 12)
    held = {}
    for k, n48 in enumerate((14403, 12000, 17999)):
        x = synth.synth_clip(95 + k, 'speech', n48)
        name = f'p257_{k:03d}.wav'
        wavio.write_wav_pcm16(os.path.join(mix, name), x, 48000)
        held[name] = wavio.read_wav(os.path.join(mix, name))[0]
    args = types.SimpleNamespace(mix_file_path=mix, esti_clean_file_path=out, fs=16000)
    st = {}
    assert decode.enhance(args, 'crn', state_dict=sd, max_batch=4, verbose=False, stats=st) == 3 and st['raw_pcm16']
    for name, x48 in held.items():
        y, fs = wavio.read_wav(os.path.join(out, name))
        ref = D.enhance_crn(sd, R.librosa_resample(x48, 48000, 16000))
        assert fs == 16000 and len(y) == len(ref)
        diff = np.abs(_pcm16(ref) - np.round(y * 32768.0).astype(np.int64))
        assert diff.max() <= 1 and rms(y - ref) < 1e-4, (name, diff.max(), rms(y - ref))
