"""Deterministic synthetic weights and clips (no datasets / checkpoints here).

47 of the reference's 51 checkpoints are stripped (`.MISSING_LARGE_BLOBS`), so
every model except DPCRN is exercised with build-generated weights that are a
pure function of (state-dict schema, seed) - identical for the reference
module (golden generation), the numpy oracle and the HIP engine.  Clips follow
SURVEY.md section 8(d) "Synthetic inputs".
"""
from collections import OrderedDict
import numpy as np


def synth_state_dict(schema, seed=0):
    """schema: OrderedDict name -> (shape tuple, 'f32' | 'i64').
    Values depend only on (name order, shapes, seed)."""
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for name, (shape, dt) in schema.items():
        shape = tuple(shape)
        leaf = name.rsplit('.', 1)[-1]
        if dt == 'i64':
            sd[name] = np.zeros(shape, dtype=np.int64)
            continue
        if leaf == 'running_var':
            v = rng.uniform(0.5, 1.5, shape)
        elif leaf == 'running_mean':
            v = 0.1 * rng.standard_normal(shape)
        elif len(shape) <= 1 and leaf.startswith('weight'):
            # norm gains; scalar / per-channel PReLU slopes
            v = rng.uniform(0.1, 0.4, shape) if shape in ((1,), ()) else rng.uniform(0.5, 1.5, shape)
        elif leaf == 'gain':
            v = rng.uniform(0.5, 1.5, shape)                  # CumulativeLayerNorm gain [1,C,1(,1)]
        elif len(shape) <= 1 or (leaf == 'bias' and shape[0] == 1):
            v = rng.uniform(-0.1, 0.1, shape)                 # biases (cLN bias is [1,C,1(,1)])
        else:
            fan_in = int(np.prod(shape[1:]))
            bound = 1.0 / np.sqrt(fan_in)
            v = rng.uniform(-bound, bound, shape)
        sd[name] = v.astype(np.float32)
    return sd


def schema_of(sd):
    """Schema (shape, dtype tag) of a state dict of numpy arrays / torch tensors."""
    out = OrderedDict()
    for k, v in sd.items():
        a = v.detach().cpu().numpy() if hasattr(v, 'detach') else np.asarray(v)
        out[k] = (tuple(a.shape), 'i64' if a.dtype == np.int64 else 'f32')
    return out


def synth_clip(seed, kind='speech', length=64000, fs=16000):
    """'speech' = 19-harmonic tone, 1/k roll-off, 3 Hz AM, peak 0.1,
    plus 0.02 N(0,1); 'white' = 0.05 N(0,1); 'quiet' = 1e-4 N(0,1);
    'gap' = speech with a 1 s all-zero gap."""
    rng = np.random.default_rng(1000 + seed)
    t = np.arange(length) / fs
    if kind in ('speech', 'gap'):
        f0 = rng.uniform(90.0, 250.0)
        x = np.zeros(length)
        for k in range(1, 20):
            x += np.sin(2 * np.pi * k * f0 * t + rng.uniform(0, 2 * np.pi)) / k
        x *= (0.5 + 0.5 * np.sin(2 * np.pi * 3.0 * t)) ** 2
        x *= 0.1 / np.max(np.abs(x))
        x += 0.02 * rng.standard_normal(length)
        if kind == 'gap':
            s = length // 3
            x[s:s + min(fs, length // 4)] = 0.0
    elif kind == 'white':
        x = 0.05 * rng.standard_normal(length)
    elif kind == 'quiet':
        x = 1e-4 * rng.standard_normal(length)
    else:
        raise ValueError(kind)
    return x.astype(np.float32)


def synth_batch(batch, kind='speech', length=64000, seed0=0):
    return np.stack([synth_clip(seed0 + b, kind, length) for b in range(batch)])
