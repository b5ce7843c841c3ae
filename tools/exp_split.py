"""Experiment (round 6): does a batch pipeline as two half-batches on two streams?  Two engines of B / 2 clips each, their decodes
enqueued on two torch streams from one thread, against one engine of B clips.  Prints utt/s for both forms.
Usage: python tools/exp_split.py --model crn --batch 64 [--parts 2]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import se_amd  # noqa: E402,F401
from se_amd import synth  # noqa: E402


def build(name, B, L):
    from se_amd import models_new
    from se_amd.models import MODEL_CLASSES
    if name == 'ctsnet_new':
        return models_new.CTSNet(max_batch=B, max_samples=L).load_synthetic(17, 18)
    if name.endswith('_new'):
        return getattr(models_new, {'g2net_new': 'G2Net', 'taylorsenet_new': 'TaylorSENet'}[name])(max_batch=B, max_samples=L).load_synthetic(1)
    return MODEL_CLASSES[name](max_batch=B, max_samples=L).load_synthetic(1)


def timed(fn, steps):
    import torch
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='crn')
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--parts', type=int, default=2)
    ap.add_argument('--steps', type=int, default=10)
    a = ap.parse_args()
    L = 64000
    B, P = a.batch, a.parts
    base = synth.synth_batch(16, 'speech', L, seed0=300)
    wav = torch.from_numpy(np.tile(base, ((B + 15) // 16, 1))[:B].copy()).cuda()
    whole = build(a.model, B, L).engine
    out = torch.empty((B, whole.output_samples(L)), dtype=torch.float32, device='cuda')
    t_whole = timed(lambda: whole.enhance_batch(wav, out), a.steps)
    del whole
    parts = [build(a.model, B // P, L).engine for _ in range(P)]
    streams = [torch.cuda.Stream() for _ in range(P)]
    outs = [torch.empty((B // P, parts[0].output_samples(L)), dtype=torch.float32, device='cuda') for _ in range(P)]
    ins = [wav[i * (B // P):(i + 1) * (B // P)].contiguous() for i in range(P)]

    def split():
        for i in range(P):
            with torch.cuda.stream(streams[i]):
                parts[i].enhance_batch(ins[i], outs[i])
    t_split = timed(split, a.steps)
    print(f"{a.model} B={B}: one engine {B / t_whole:.1f} utt/s ({t_whole * 1e3:.2f} ms), {P} engines of {B // P} on {P} streams "
          f"{B / t_split:.1f} utt/s ({t_split * 1e3:.2f} ms)", flush=True)


if __name__ == '__main__':
    main()
