#!/bin/bash
# full GPU suite + the driver's bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/r4_full
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r4_full/bench.json 2> gpurun_out/r4_full/bench.err
python - <<'PY'
import json,os
d=json.load(open(os.path.join(os.environ.get('GRAFT_REPO_ROOT','/root/repo'),'gpurun_out/r4_full/bench.json')))
print('value',d['value'],'frac',d['roofline']['frac'],'whole',d['roofline_whole_path']['frac'])
for r in d['roofline'].get('configs',[])+d['roofline'].get('zoo',[]): print(r['model'],r['batch'],r['utt_s'],r['frac'])
print('cpu',d.get('cpu_baseline',{}).get('value'))
for s in d.get('roofline_stages',[]): print(s['stage'],s['ms_per_step'],s['frac'])
PY
