#!/usr/bin/env python
"""Parity-margin record: how much of the 1e-4 RMS bar every network uses, per fixture.

  python tools/parity_record.py [--out gpurun_out/r05_parity.json] [--models a,b] [--no-batch] [--no-long]
  SE_ENGINE_LIB=<pkg>/libse_engine_exact.so python tools/parity_record.py --out gpurun_out/r05_parity_exact.json

Per network and fixture (all produced by the IMPORTED reference, oracle/gen_golden.py; DCCRN on the complexnn recall):
  alone4   the 4 s fixture clip in a batch of 3                              (tests/test_gpu_full_fixture.py)
  row256   the same clip as one row of the sweep batch (256; FullSubNet 128) (tests/test_gpu_b256_fixture.py)
  long10 / long15   the 160 000- / 239 987-sample clips alone                (tests/test_gpu_long_clips.py)
-> rms_err (engine - reference waveform), rms_ref, used = rms_err / 1e-4 (the absolute bar), used_rel = rms_err / (5e-4 *
max(rms_ref, 1e-3)) (the relative bar the tests also assert).  Plus the PESQ / STOI deltas of tests/test_metrics.py.
The same script under SE_ENGINE_LIB=libse_engine_exact.so (make -C csrc exact: libm transcendentals, fastmath.h) gives the
cost of the hardware approximations as a number.  Reads only tests/golden (nothing under /root/reference).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402


def rms(a):
    a = np.asarray(a, np.float64)
    return float(np.sqrt(np.mean(a * a)))


def entry(got, ref):
    e, r = rms(got - ref), rms(ref)
    return {'rms_err': e, 'rms_ref': r, 'used': e / 1e-4, 'used_rel': e / (5e-4 * max(r, 1e-3)), 'finite': bool(np.isfinite(got).all())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'r05_parity.json'))
    ap.add_argument('--models', default='')
    ap.add_argument('--no-batch', action='store_true')
    ap.add_argument('--no-long', action='store_true')
    args = ap.parse_args()
    import torch
    import se_amd  # noqa: F401
    from se_amd import synth, _lib
    from conftest import load_golden
    from test_gpu_b256_fixture import SEEDS, BATCH, make, fixture_clip, L
    names = [n for n in sorted(SEEDS) if not args.models or n in args.models.split(',')]
    rec = {'lib': os.path.basename(_lib.LIB_PATH), 'bar_abs': 1e-4, 'bar_rel': '5e-4 * max(rms_ref, 1e-3)', 'models': {}}
    for name in names:
        out = {}
        clip, ref = fixture_clip(name)
        x = np.stack([synth.synth_clip(900, 'white', L), clip, synth.synth_clip(901, 'speech', L)])
        m = make(name, 3)
        out['alone4'] = entry(m.enhance_batch(torch.from_numpy(x).cuda()).cpu().numpy()[1], ref)
        del m
        if not args.no_batch:
            B = BATCH.get(name, 256)
            base = synth.synth_batch(16, 'speech', L, seed0=700)
            xb = np.tile(base, ((B + 15) // 16, 1))[:B].copy()
            row = 5 + 16 * ((B // 16) // 2)
            xb[row] = clip
            m = make(name, B)
            out['row%d' % B] = entry(m.enhance_batch(torch.from_numpy(xb).cuda())[row].cpu().numpy(), ref)
            del m
        if not args.no_long:
            G15 = load_golden('long15_' + name)
            m = make(name, 1, int(G15['n']))
            for tag in ('10', '15'):
                G = load_golden('long%s_%s' % (tag, name))
                c = synth.synth_clip(int(G['seed']), 'speech', int(G['n']))
                out['long' + tag] = entry(m.enhance_batch(torch.from_numpy(c[None]).cuda()).cpu().numpy()[0], G['enh_cprs'])
            del m
        torch.cuda.empty_cache()
        rec['models'][name] = out
        print(name, json.dumps({k: round(v['used'], 6) for k, v in out.items()}), flush=True)
    # quality gate of BASELINE (PESQ within +-0.01, STOI to 3 d.p.): the two GPU checks of tests/test_metrics.py as numbers
    try:
        from se_amd import pesq as P, metrics
        from se_amd.models import dpcrn, crn_net
        from oracle import decode as D
        import test_metrics as TM
        clean = TM._utterances(7)
        noisy = (clean + 0.02 * np.random.default_rng(8).standard_normal(len(clean))).astype(np.float32)
        ck = dict(load_golden('ckpt_vb_dpcrn_noncprs'))
        m = dpcrn(max_batch=1, max_samples=len(noisy))
        m.load_state_dict(ck)
        y = m.enhance_batch(torch.from_numpy(noisy[None]).cuda()).cpu().numpy()[0].astype(np.float64)
        refw = D.enhance_dpcrn(ck, noisy.astype(np.float64))
        p_eng, p_ref = P.pesq(clean, y), P.pesq(clean, refw)
        rec['pesq_dpcrn_real_ckpt'] = {'engine': p_eng, 'reference_path': p_ref, 'delta': p_eng - p_ref, 'gate': 0.01}
        clean = TM._speechlike(5)
        noisy = (clean + 0.03 * np.random.default_rng(6).standard_normal(len(clean))).astype(np.float32)
        m = crn_net(max_batch=1, max_samples=len(noisy)).load_synthetic(12)
        sd = synth.synth_state_dict(m.state_dict_schema(), 12)
        y = m.enhance_batch(torch.from_numpy(noisy[None]).cuda()).cpu().numpy()[0].astype(np.float64)
        refw = D.enhance_crn(sd, noisy.astype(np.float64))
        rec['stoi_crn'] = {'engine': metrics.stoi(clean, y, 16000), 'reference_path': metrics.stoi(clean, refw, 16000)}
        rec['stoi_crn']['delta'] = rec['stoi_crn']['engine'] - rec['stoi_crn']['reference_path']
    except Exception as e:      # the scorers are off the hot path: record the failure, keep the table
        rec['metrics_error'] = repr(e)
    worst = max(((v['used'], n, k) for n, o in rec['models'].items() for k, v in o.items()), default=(0, '', ''))
    worst_rel = max(((v['used_rel'], n, k) for n, o in rec['models'].items() for k, v in o.items()), default=(0, '', ''))
    rec['worst_abs'] = {'used': worst[0], 'model': worst[1], 'fixture': worst[2]}
    rec['worst_rel'] = {'used_rel': worst_rel[0], 'model': worst_rel[1], 'fixture': worst_rel[2]}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, 'w') as f:
        json.dump(rec, f, indent=1)
    print('worst share of the 1e-4 bar: %.4g (%s, %s); of the relative bar: %.4g (%s, %s)' % (worst + worst_rel))


if __name__ == '__main__':
    main()
