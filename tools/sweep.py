#!/usr/bin/env python
"""Throughput survey over every model the engine builds (not the headline bench): whole decode path on 4 s clips.

  python tools/sweep.py [--batch 64] [--steps 3] [--models dccrn,crn,...]
Prints one JSON line per model: utt/s, ms/step, share of the step spent in the tap-table GEMM family.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

GFLOP = {'lstm': 17.5, 'crn': 17.3, 'gcrn': 13.2, 'dpcrn': 4.8, 'dccrn': 53.4, 'fullsubnet': 238.5, 'ctsnet': 25.6,
         'g2net': 10.7, 'taylorsenet': 31.3, 'uformer': 27.5}      # SURVEY 8(d), per 4 s utterance


SAMPLES = 64000


def build(name, B):
    import se_amd  # noqa: F401
    from se_amd import models, models_new
    kw = dict(max_batch=B, max_samples=SAMPLES)
    if name == 'ctsnet':
        return models.CTSNet(**kw).load_synthetic(17, 18)
    if name == 'ctsnet_new':
        return models_new.CTSNet(**kw).load_synthetic(17, 18)
    return models.MODEL_CLASSES[name](**kw).load_synthetic(1)


def ragged_pass(name, B, n_clips):
    """Real-corpus shape: n_clips clips, every one with its own length, decoded through se_amd.decode.plan_batches
    (sorted runs of <= B clips within B x 4 s of padded samples) and se_enhance_ragged.  Reports clips/s, audio seconds
    per second and the padding the calls carried."""
    import torch
    from se_amd import synth
    from se_amd.decode import plan_batches, RAGGED_MODELS
    rng = np.random.default_rng(2024)
    secs = np.clip(np.exp(rng.normal(np.log(2.6), 0.45, 4 * n_clips)), 1.2, 9.8)
    lengths = sorted(set(int(v * 16000) for v in secs))
    rng.shuffle(lengths)
    lengths = lengths[:n_clips]
    ragged = name in RAGGED_MODELS
    batches = plan_batches(lengths, B, B * 64000, ragged)
    eb, el = max(len(b) for b in batches), max(lengths)
    import se_amd  # noqa: F401
    from se_amd import models, models_new
    kw = dict(max_batch=eb, max_samples=el)
    if name.startswith('ctsnet'):
        m = (models_new if name.endswith('_new') else models).CTSNet(**kw).load_synthetic(17, 18)
    else:
        m = models.MODEL_CLASSES[name](**kw).load_synthetic(1)
    base = torch.from_numpy(synth.synth_clip(5, 'speech', el)).cuda()
    calls = []
    for b in batches:
        lens = [lengths[i] for i in b]
        calls.append((base[None, :max(lens)].repeat(len(b), 1).contiguous(), lens))

    def run():
        for wav, lens in calls:
            if min(lens) == max(lens):
                m.enhance_batch(wav)
            else:
                m.enhance_ragged(wav, lens)
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    audio = sum(lengths) / 16000.0
    padded = sum(len(b) * max(lengths[i] for i in b) for b in batches) / 16000.0
    return {'ragged_clips': len(lengths), 'ragged_calls': len(batches), 'ragged_utt_per_s': round(len(lengths) / dt, 1),
            'ragged_x_realtime': round(audio / dt, 0), 'ragged_pad_frac': round(padded / audio - 1.0, 4),
            'ragged_mean_clip_s': round(audio / len(lengths), 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--no-profile', action='store_true', help='skip the HIP-event kernel timing (needed for SE_GRAPH=1 replay)')
    ap.add_argument('--ragged', type=int, default=0, metavar='N',
                    help='also decode N clips of N distinct lengths (VoiceBank+DEMAND-like: 1.2 - 9.8 s, median ~2.5 s) '
                         'through the driver\'s batch plan + se_enhance_ragged')
    ap.add_argument('--models', type=str, default='lstm,crn,gcrn,dpcrn,dccrn,fullsubnet,ctsnet,g2net,taylorsenet,uformer')
    ap.add_argument('--samples', type=int, default=64000, help='clip length (64 000 = 4 s: 401 frames at hop 160; 66 400: 416 frames)')
    ap.add_argument('--fsn-max-batch', type=int, default=128, help='largest FullSubNet batch (batch sweeps around 128 raise it)')
    args = ap.parse_args()
    global SAMPLES
    SAMPLES = args.samples
    import torch
    from se_amd import synth
    B = args.batch
    base = synth.synth_batch(8, 'speech', args.samples, seed0=100)
    wav = torch.from_numpy(np.tile(base, ((B + 7) // 8, 1))[:B].copy()).cuda()
    B0 = B
    for name in args.models.split(','):
        B = min(B0, args.fsn_max_batch) if name == 'fullsubnet' else B0       # 257 * B sub-band sequences: 256 clips do not fit 288 GB
        if B != wav.shape[0]:
            wav = torch.from_numpy(np.tile(base, ((B + 7) // 8, 1))[:B].copy()).cuda()
        m = build(name, B)
        eng = m.engine
        out = torch.empty((B, eng.output_samples(args.samples)), dtype=torch.float32, device='cuda')
        eng.enhance_batch(wav, out)
        torch.cuda.synchronize()
        eng.enhance_batch(wav, out)          # second call of the shape (captures the graph when SE_GRAPH=1)
        torch.cuda.synchronize()
        # throughput first, unprofiled (the per-launch HIP events of the profiler cost up to 8 % on the models with
        # hundreds of small launches per step); then one profiled pass for the tap-table GEMM family's own time
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.enhance_batch(wav, out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        prof = {'gemm_ms': 0.0, 'gemm_launches': 0, 'gemm_flops': 0.0}
        if not args.no_profile:
            eng.set_profiling(True)
            eng.enhance_batch(wav, out)
            torch.cuda.synchronize()
            prof = eng.get_profile()
            eng.set_profiling(False)
        assert bool(torch.isfinite(out).all()), name
        ups = B / dt
        g = GFLOP.get(name.replace('_new', ''), 0.0)
        row = {'model': name, 'batch': B, 'utt_per_s': round(ups, 1), 'ms_per_step': round(dt * 1e3, 2),
               'x_realtime': round(ups * 4, 0), 'algo_tflops': round(ups * g / 1e3, 2),
               'gemm_ms': round(prof['gemm_ms'], 2), 'gemm_launches': prof['gemm_launches'],
               'gemm_tflops': round(prof['gemm_flops'] / max(prof['gemm_ms'], 1e-9) / 1e9, 2)}
        del m, eng
        if args.ragged:
            row.update(ragged_pass(name, B, args.ragged))
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
