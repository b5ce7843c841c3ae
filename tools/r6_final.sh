#!/bin/bash
# round-6 evidence run (through gpurun, two calls: `bash tools/r6_final.sh a`, `... b`) -> gpurun_out/r6/ and gpurun_out/r06_*
#   a: the driver's bench line + kernel summary + the three DCCRN PMC passes (tools/measure_round.sh), parity-margin record,
#      PMC passes of TaylorSENet / G2Net / FullSubNet / Uformer
#   b: kernel summaries of the zoo (multi-stream models - and the models that decode two half-batches side by side - on ONE stream, so that the per-launch durations are not
#      stretched by concurrent kernels), sweeps at batch 1 / 64 / 256, frame-online latencies, the file -> file corpus bench
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6
mkdir -p $OUT
PART=${1:-a}
if [ "$PART" = a ]; then
  bash $ROOT/tools/measure_round.sh r06 2>&1 | tail -3 | cut -c1-300
  bash $ROOT/tools/r6_call.sh parity
  SE_BATCH_SPLIT=0 bash $ROOT/tools/pmc_models.sh r06 "taylorsenet 256" "fullsubnet 128" 2>&1 | grep -v "^declare" | tail -4 | cut -c1-200
  SE_BATCH_SPLIT=0 bash $ROOT/tools/pmc_models.sh r06 "uformer 256" 2>&1 | grep -v "^declare" | tail -2 | cut -c1-200      # (one stream: per-kernel busy cycles)
  SE_G2NET_FORK=0 bash $ROOT/tools/pmc_models.sh r06 "g2net 256" 2>&1 | grep -v "^declare" | tail -2 | cut -c1-200      # (one stream: per-kernel busy cycles)
else
  export SE_TAYLOR_FORK=0 SE_FSN_SPLIT=1 SE_BATCH_SPLIT=0
  bash $ROOT/tools/r6_call.sh prof:taylorsenet:256 prof:taylorsenet_new:256 prof:fullsubnet:128 2>&1 | grep utt_per_s | cut -c1-90
  unset SE_TAYLOR_FORK SE_FSN_SPLIT SE_BATCH_SPLIT
  export SE_G2NET_FORK=0
  bash $ROOT/tools/r6_call.sh prof:g2net:256 prof:g2net_new:256 2>&1 | grep utt_per_s | cut -c1-90
  unset SE_G2NET_FORK
  export SE_BATCH_SPLIT=0
  bash $ROOT/tools/r6_call.sh prof:uformer:256 prof:dpcrn:256 prof:ctsnet:256 prof:ctsnet_new:256 2>&1 | grep utt_per_s | cut -c1-90
  unset SE_BATCH_SPLIT
  bash $ROOT/tools/r6_call.sh prof:gcrn:256 prof:crn:64 2>&1 | grep utt_per_s | cut -c1-90
  cd $ROOT
  ALL=lstm,crn,gcrn,dpcrn,dccrn,fullsubnet,ctsnet,g2net,taylorsenet,uformer,ctsnet_new,g2net_new,taylorsenet_new
  for B in 1 64 256; do
    timeout 600 python tools/sweep.py --batch $B --steps 5 --models $ALL > $OUT/r06_sweep_b$B.jsonl 2> $OUT/sweep_b$B.err
    wc -l $OUT/r06_sweep_b$B.jsonl
  done
  timeout 300 python tools/stream_latency.py > $OUT/r06_stream_latency.jsonl 2> $OUT/stream.err
  wc -l $OUT/r06_stream_latency.jsonl
  timeout 600 python tools/corpus_bench.py > $OUT/r06_corpus.json 2> $OUT/corpus.err
  tail -c 400 $OUT/r06_corpus.json
fi
