#!/bin/bash
# the micro-benchmarks of DESIGN.md 3.2 "Round 4" + the frame-online push latencies -> gpurun_out/r04_micro/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04_micro
mkdir -p $OUT
cd $ROOT/sixty*/
timeout 120 ./mfma4bench > $OUT/mfma4bench.log 2>&1
timeout 120 ./coissuebench > $OUT/coissuebench.log 2>&1
{
for spec in "1024 64 401 1" "1024 128 401 1" "1024 256 401 1" "512 256 401 2" "1024 64 401 2"; do
  echo "== $spec: default (lstm_coop16_kernel from 17 sequences on)"; timeout 120 ./coopbench $spec
  echo "== $spec: SE_COOP16=0 SE_COOP4=0 (round-3 flag-exchange kernel)"; SE_COOP16=0 SE_COOP4=0 timeout 120 ./coopbench $spec
done
} > $OUT/coopbench.log 2>&1
cd $ROOT
timeout 600 python tools/stream_latency.py > $OUT/stream_latency.jsonl 2> $OUT/stream_latency.err
tail -3 $OUT/mfma4bench.log; tail -3 $OUT/coissuebench.log; grep "coop LSTM" $OUT/coopbench.log | cut -c1-120; cat $OUT/stream_latency.jsonl | cut -c1-140
