#!/bin/bash
# round 4, GPU call 2: the 8-wave sub-tile cooperative LSTM against the round-3 kernel, the recurrent models' sweep
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4_2
mkdir -p $OUT
cd $ROOT
P=sixty-years-of-frequency-domain-monaural-speech-enhancement_amd
export COOPBENCH_TC=12
{
for cfg in "1024 64 401 1" "1024 256 401 1" "1024 128 401 1" "1024 32 401 1" "1024 16 401 1" "1024 8 401 1" "1024 70 401 1" "512 128 251 1" "512 256 401 2" "512 64 401 2"; do
  echo "== $cfg: round-3 kernel"; SE_COOP4=0 timeout 90 $P/coopbench $cfg
  for L in 1 2 3; do echo "== $cfg: coop8 lead $L"; SE_COOP4_MINS=5 SE_COOP4_LEAD=$L timeout 90 $P/coopbench $cfg; done
  echo "== $cfg: coop8, no tag check (wrong results: ablation)"; SE_COOP4_MINS=5 SE_COOP_DBG=4 timeout 90 $P/coopbench $cfg
done
} > $OUT/coopbench.log 2>&1
grep -E "^==|us/step|max" $OUT/coopbench.log | paste - - - | awk '{print $2,$3,$4,$5,$6,$7,$8,$9, "|", $0}' | sed -E 's/\|.*max \|gpu - float64 host\| over 6 sequences x 12 steps: ([0-9.e+-]+).*T=[0-9]+ Z=[0-9]: ([0-9.]+) ms +([0-9.]+) us\/step.*/| maxdiff \1 | \3 us\/step/' | head -60
for B in 64 256; do timeout 600 python tools/sweep.py --models lstm,crn,gcrn --batch $B --steps 5 --no-profile 2>&1 | grep utt_per_s | cut -c1-100; done
timeout 300 python tools/sweep.py --models fullsubnet --batch 128 --steps 3 --no-profile 2>&1 | grep utt_per_s | cut -c1-100
