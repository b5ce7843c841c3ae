#!/bin/bash
# round 4, GPU call 1: the sub-tile pipelined cooperative LSTM against the round-3 kernel (tools/coopbench.cpp), the recurrent
# models' parity tests on it, the bench line with all 13 networks
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4_1
mkdir -p $OUT
cd $ROOT
P=sixty-years-of-frequency-domain-monaural-speech-enhancement_amd
export COOPBENCH_TC=12
{
for cfg in "1024 64 401 1" "1024 256 401 1" "1024 128 401 1" "1024 32 401 1" "1024 16 401 1" "1024 70 401 1" "512 128 251 1" "512 256 401 2" "512 64 401 2"; do
  echo "== $cfg: round-3 kernel"; SE_COOP4=0 timeout 90 $P/coopbench $cfg
  for L in 1 2 3; do echo "== $cfg: coop4 dma lead $L"; SE_COOP4_MINS=5 SE_COOP4_LEAD=$L timeout 90 $P/coopbench $cfg; done
  echo "== $cfg: coop4 dma default lead"; SE_COOP4_MINS=5 timeout 90 $P/coopbench $cfg
  echo "== $cfg: coop4 registers lead 2"; SE_COOP4_MINS=5 SE_COOP4_DMA=0 SE_COOP4_LEAD=2 timeout 90 $P/coopbench $cfg
  echo "== $cfg: coop4 dma, no tag check (wrong results: ablation)"; SE_COOP4_MINS=5 SE_COOP_DBG=4 timeout 90 $P/coopbench $cfg
done
} > $OUT/coopbench.log 2>&1
grep -E "^==|us/step|max" $OUT/coopbench.log | head -150
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_full_fixture.py tests/test_gpu_edge_cases.py tests/test_gpu_decode_driver.py tests/test_gpu_streaming.py -x -q -m gpu 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_b256_fixture.py tests/test_gpu_long_clips.py tests/test_gpu_full_size.py tests/test_gpu_ragged.py -x -q -m gpu -k "lstm or crn or gcrn or fullsubnet" 2>&1 | tail -5
timeout 600 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json,os
d=json.load(open(os.path.join(os.environ.get('GRAFT_REPO_ROOT','/root/repo'),'gpurun_out/r4_1/bench.json')))
print('value',d['value'],'frac',d['roofline']['frac'])
for r in d['roofline'].get('configs',[])+d['roofline'].get('zoo',[]): print(r['model'],r['batch'],r['utt_s'],r['frac'])
print('cpu',d.get('cpu_baseline',{}).get('value'))
PY
