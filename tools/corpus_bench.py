#!/usr/bin/env python
"""File -> file throughput of the decode driver (se_amd/decode.py:enhance) on a synthetic VoiceBank+DEMAND-shaped corpus.

Writes `--files` PCM_16 clips (default 824 - the size of the VoiceBank+DEMAND test set - at 48 kHz, every clip with its own
length, 1.2 - 9.8 s, mean ~3 s) to a tmpfs directory, decodes the directory through the driver (header scan -> plan ->
reader threads + side-stream upload / PCM decode / 48 -> 16 kHz resampling -> se_enhance_ragged -> device-side PCM_16 ->
writer thread), and prints one JSON line: clips/s and x real time of the pipeline, the padding its calls carried, and - for
comparison - the same calls decoded from device-resident tensors (what tools/sweep.py --ragged reports).  The DCCRN decode
script it stands in for: DCCRN/dccrn_decode_vb.py:22-64.

    python tools/corpus_bench.py [--model dccrn] [--files 824] [--fs 48000] [--max_batch 64] [--dir /dev/shm/se_corpus]
"""
import argparse
import json
import os
import shutil
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def corpus_lengths(n, seed=2024):
    rng = np.random.default_rng(seed)
    secs = np.clip(np.exp(rng.normal(np.log(2.6), 0.45, 4 * n)), 1.2, 9.8)
    lengths = sorted(set(int(v * 16000) for v in secs))
    rng.shuffle(lengths)
    return lengths[:n]


def write_corpus(d, n, fs):
    """n clips cut from one long speech-like signal (cheap to make), each with its own gain, PCM_16 at `fs`."""
    from se_amd import synth, wavio
    os.makedirs(d, exist_ok=True)
    lens16 = corpus_lengths(n)
    r = fs // 16000
    base = synth.synth_clip(77, 'speech', 10 * fs + 4096)
    for k, n16 in enumerate(lens16):
        off = (k * 977) % 4096
        wavio.write_wav_pcm16(os.path.join(d, f'p{232 + k % 28}_{k:04d}.wav'), base[off:off + n16 * r] * (0.4 + 0.05 * (k % 9)), fs)
    return lens16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='dccrn')
    ap.add_argument('--files', type=int, default=824)
    ap.add_argument('--fs', type=int, default=48000)
    ap.add_argument('--max_batch', type=int, default=64)
    ap.add_argument('--readers', type=int, default=4)
    ap.add_argument('--dir', default='/dev/shm/se_corpus')
    ap.add_argument('--keep', action='store_true')
    ap.add_argument('--repeat', type=int, default=2, help='at least this many passes over the directory')
    ap.add_argument('--min_seconds', type=float, default=10.0, help='keep decoding the directory until the passes after the '
                    'first add up to this much pipeline time; the figure reported is their aggregate (the first pass also '
                    'pays one-time costs - code-object load, resampler tables, first touch of pinned pages - and is reported apart)')
    args = ap.parse_args()
    import torch
    import se_amd  # noqa: F401
    from se_amd import decode, synth, schemas, wavio
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    if 'LOCAL_RANK' in os.environ:
        torch.cuda.set_device(int(os.environ['LOCAL_RANK']) % torch.cuda.device_count())
    mix, out = os.path.join(args.dir, 'noisy'), os.path.join(args.dir, 'enh')
    if rank == 0:
        shutil.rmtree(args.dir, ignore_errors=True)
        t0 = time.perf_counter()
        write_corpus(mix, args.files, args.fs)
        t_corpus = time.perf_counter() - t0
        open(os.path.join(args.dir, 'ready'), 'w').close()
    else:
        t_corpus = 0.0
        while not os.path.exists(os.path.join(args.dir, 'ready')):
            time.sleep(0.05)
    name = args.model
    if name.startswith('ctsnet'):
        tag = '_new' if name.endswith('_new') else ''
        sd = (synth.synth_state_dict(schemas.SCHEMAS['cts_step1' + tag](), 17), synth.synth_state_dict(schemas.SCHEMAS['cts_step2' + tag](), 18))
    else:
        sd = synth.synth_state_dict(schemas.SCHEMAS[name](), 1)
    ns = types.SimpleNamespace(mix_file_path=mix, esti_clean_file_path=out, fs=16000)
    first, passes, t_pipe, n_clips, per_pass = None, 0, 0.0, 0, []
    while passes < max(2, args.repeat) or (t_pipe < args.min_seconds and passes < 400):
        stats = {}
        n = decode.enhance(ns, name, state_dict=sd, max_batch=args.max_batch, p_in=0.5, p_out=2.0, verbose=False, stats=stats,
                           readers=args.readers, rank=rank, world=world)
        passes += 1
        if first is None:
            first = dict(stats)
            continue
        t_pipe += stats['pipeline_s']
        n_clips += n
        per_pass.append(stats['clips_per_s'])
    torch.cuda.synchronize()
    stats['passes_timed'], stats['pipeline_s'] = passes - 1, round(t_pipe, 3)
    stats['clips_per_s'] = round(n_clips / t_pipe, 1)
    stats['clips_per_s_min_max'] = [min(per_pass), max(per_pass)]
    stats['audio_s_rank'] = round(stats['audio_s_rank'] * (passes - 1), 2)
    # ---- the same calls from device-resident tensors (no file I/O, no upload, no resampling, no PCM conversion)
    lens16 = [wavio.wav_info(os.path.join(mix, f))[0] * 16000 // args.fs for f in sorted(os.listdir(mix))]
    own = decode.shard_clips(lens16, rank, world)
    plan = decode.plan_batches([lens16[i] for i in own], args.max_batch, args.max_batch * 64000, True)
    eb, el = max(len(b) for b in plan), max(lens16[i] for i in own)
    net = decode._build(name, None, sd, max_batch=eb, max_samples=el, p_in=0.5, p_out=2.0)
    base = torch.from_numpy(synth.synth_clip(5, 'speech', el)).cuda()
    calls = []
    for b in plan:
        lens = [lens16[own[k]] for k in b]
        calls.append((base[None, :max(lens)].repeat(len(b), 1).contiguous(), lens))

    def run():
        for wav, lens in calls:
            net.enhance_batch(wav) if min(lens) == max(lens) else net.enhance_ragged(wav, lens)
    run()
    torch.cuda.synchronize()
    reps = max(3, min(200, int(0.3 * args.min_seconds / max(t_pipe / max(passes - 1, 1), 1e-3))))
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    row = {'tool': 'corpus_bench', 'model': name, 'fs_in': args.fs, 'max_batch': args.max_batch, 'rank': rank, **stats,
           'x_realtime': round(stats['audio_s_rank'] / stats['pipeline_s'], 0),
           'resident_clips_per_s': round(len(own) / dt, 1),
           'file_vs_resident': round(stats['clips_per_s'] / (len(own) / dt), 3),
           'first_pass': {k: first[k] for k in ('setup_s', 'pipeline_s', 'clips_per_s')},
           'corpus_write_s': round(t_corpus, 2), 'host_cpus': os.cpu_count()}
    assert n == len(own) and len(os.listdir(out)) >= n
    print(json.dumps(row), flush=True)
    if rank == 0 and not args.keep and world == 1:
        shutil.rmtree(args.dir, ignore_errors=True)


if __name__ == '__main__':
    main()
