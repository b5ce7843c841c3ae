"""Determinism check: repeated decodes of one batch must be bit-identical (eager and hipGraph replay)."""
import sys
import os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import se_amd  # noqa: F401
from se_amd import synth
from se_amd.models import MODEL_CLASSES
name = sys.argv[1]
seeds = dict(crn=12, dccrn=14, g2net=20, lstm=11, dpcrn=13, fullsubnet=15, gcrn=16, taylorsenet=19)
x = np.stack([synth.synth_clip(90 + b, 'speech', 4000) for b in range(2)])
for graphs in (False, True):
    m = MODEL_CLASSES[name](max_batch=2, max_samples=4000, graphs=graphs).load_synthetic(seeds[name])
    ys = [m.enhance_batch(torch.from_numpy(x).cuda()).cpu().numpy() for _ in range(5)]
    print(name, 'graphs' if graphs else 'eager', [float(np.abs(y - ys[0]).max()) for y in ys])
