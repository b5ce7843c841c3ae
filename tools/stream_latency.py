"""Frame-online decode cost per model: wall time of one se_stream_push that completes `chunk` STFT frames, for B parallel
streams (synthetic weights, seeded clips).  Prints one JSON line per (model, B, chunk):
  ms_per_push, frames per push, x_realtime = audio time of the push / its wall time (per stream), host_enqueue_ms = the part of it the
  call itself takes (everything enqueued, nothing waited for); launches are not counted.
Usage: python tools/stream_latency.py [--models crn,dccrn,ctsnet_new] [--batch 1,16] [--chunk 1,8]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import se_amd  # noqa: E402,F401
from se_amd import synth  # noqa: E402

SEEDS = {'crn': 12, 'lstm': 11, 'gcrn': 16, 'dpcrn': 13, 'dccrn': 14, 'taylorsenet_new': 19, 'g2net_new': 20}


def build(name, B, L):
    from se_amd import models_new
    from se_amd.models import MODEL_CLASSES
    if name == 'ctsnet_new':
        return models_new.CTSNet(max_batch=B, max_samples=L).load_synthetic(17, 18)
    if name == 'fullsubnet_cum':        # the causal norm (base_model.py:143-166) is what makes FullSubNet streamable
        from se_amd.models import Model
        return Model(max_batch=B, max_samples=L, norm_type='cumulative_laplace_norm').load_synthetic(15)
    return MODEL_CLASSES[name](max_batch=B, max_samples=L).load_synthetic(SEEDS[name])


def main():
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument('--models', default='crn,lstm,gcrn,dpcrn,dccrn,ctsnet_new,taylorsenet_new,g2net_new,fullsubnet_cum')
    ap.add_argument('--batch', default='1,16')
    ap.add_argument('--chunk', default='1,8')
    ap.add_argument('--seconds', type=float, default=2.0)
    a = ap.parse_args()
    L = int(a.seconds * 16000)
    for name in a.models.split(','):
        for B in map(int, a.batch.split(',')):
            m = build(name, B, L)
            eng = m.engine
            hop = {'dccrn': 128, 'fullsubnet_cum': 256}.get(name, 160)
            x = torch.from_numpy(np.stack([synth.synth_clip(900 + b, 'speech', L) for b in range(B)])).cuda()
            c = eng.rms_scale(x)
            for chunk in map(int, a.chunk.split(',')):
                piece = chunk * hop
                times = []
                for rep in range(2):           # first pass warms up (lazy state slots, kernel attributes)
                    eng.stream_begin(B, c=c, max_chunk_frames=chunk)
                    times, enq = [], []
                    for p in range(0, L - piece + 1, piece):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        eng.stream_push(x[:, p:p + piece])
                        t1 = time.perf_counter()          # the call has returned: everything is enqueued
                        torch.cuda.synchronize()
                        times.append(time.perf_counter() - t0)
                        enq.append(t1 - t0)
                    eng.stream_flush()
                t = float(np.median(times[4:]))
                print(json.dumps({'model': name, 'streams': B, 'frames_per_push': chunk, 'ms_per_push': round(t * 1e3, 3),
                                  'p95_ms': round(float(np.percentile(times[4:], 95)) * 1e3, 3),
                                  'host_enqueue_ms': round(float(np.median(enq[4:])) * 1e3, 3),
                                  'x_realtime_per_stream': round(piece / 16000 / t, 2),
                                  'x_realtime_all_streams': round(B * piece / 16000 / t, 1)}), flush=True)
            del m, eng


if __name__ == '__main__':
    main()
