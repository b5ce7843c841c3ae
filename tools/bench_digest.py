#!/usr/bin/env python
"""Short digest of a bench.py JSON line: python tools/bench_digest.py gpurun_out/r5/bench.json"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get('roofline', {})
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', r.get('frac'), 'executed_frac', r.get('executed_frac'),
      'traffic', r.get('traffic'), 'stale', r.get('traffic_stale'))
for row in r.get('configs', []) + r.get('zoo', []):
    print('%-16s B %3d  %8.1f utt/s  frac %.4f  passes %s' % (row['model'], row['batch'], row['utt_s'], row['frac'], row.get('ms_per_step_passes')))
print('cpu', d.get('cpu_baseline', {}).get('value'))
for s in d.get('roofline_stages', []):
    print(s['stage'], s['ms_per_step'], s['frac'])
