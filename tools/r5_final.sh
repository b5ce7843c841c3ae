#!/bin/bash
# round-5 evidence run (through gpurun): PMC passes of G2Net / Uformer, kernel summaries of the zoo, sweeps at batch 1 / 64 / 256,
# frame-online latencies -> gpurun_out/r5/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5
mkdir -p $OUT
bash $ROOT/tools/pmc_models.sh r05 "g2net 256" "uformer 256" 2>&1 | tail -4
bash $ROOT/tools/r5_call.sh prof:g2net:256 prof:taylorsenet:256 prof:ctsnet:256 prof:uformer:256 prof:g2net_new:256 prof:taylorsenet_new:256 prof:ctsnet_new:256 prof:dpcrn:256 prof:crn:64 prof:fullsubnet:128 2>&1 | grep utt_per_s | cut -c1-90
cd $ROOT
ALL=lstm,crn,gcrn,dpcrn,dccrn,fullsubnet,ctsnet,g2net,taylorsenet,uformer,ctsnet_new,g2net_new,taylorsenet_new
for B in 1 64 256; do
  timeout 600 python tools/sweep.py --batch $B --steps 5 --models $ALL > $OUT/r05_sweep_b$B.jsonl 2> $OUT/sweep_b$B.err
  wc -l $OUT/r05_sweep_b$B.jsonl
done
timeout 300 python tools/stream_latency.py > $OUT/r05_stream_latency.jsonl 2> $OUT/stream.err
wc -l $OUT/r05_stream_latency.jsonl
