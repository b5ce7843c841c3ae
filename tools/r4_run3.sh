#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
bash tools/pmc_models.sh r04 "crn 64" "uformer 256" "g2net 256" "fullsubnet 128" 2>/dev/null | grep launches_per_step | cut -c1-200
bash tools/measure_round.sh r04 2>&1 | tail -1 | cut -c1-200
bash tools/prof_models.sh "uformer 256" "crn 64" "fullsubnet 128" "g2net 256" "dpcrn 256" 2>&1 | grep utt_per_s
for m in uformer_b256 crn_b64 fullsubnet_b128 g2net_b256 dpcrn_b256; do cp gpurun_out/r03_${m}_kernel_stats.csv gpurun_out/r04_${m}_kernel_stats.csv; done
for b in 1 64 256; do timeout 900 python tools/sweep.py --batch $b --steps 3 --models lstm,crn,gcrn,dpcrn,dccrn,fullsubnet,ctsnet,g2net,taylorsenet,uformer,ctsnet_new,g2net_new,taylorsenet_new > gpurun_out/r04_sweep_b$b.jsonl 2>/dev/null; done
timeout 300 python tools/corpus_bench.py > gpurun_out/r04_corpus.json 2>/dev/null; tail -c 300 gpurun_out/r04_corpus.json
timeout 600 python tools/stream_latency.py > gpurun_out/r04_stream_latency.jsonl 2>/dev/null; grep '"streams": 1, "frames_per_push": 1,' gpurun_out/r04_stream_latency.jsonl | cut -c1-90
