#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/r4_evid
timeout 600 python -m pytest tests/test_gpu_dccrn.py -x -q -m gpu 2>&1 | tail -5
timeout 300 python tools/corpus_bench.py > gpurun_out/r4_evid/corpus.json 2> gpurun_out/r4_evid/corpus.err; tail -c 900 gpurun_out/r4_evid/corpus.json
bash tools/pmc_models.sh r04 "crn 64" "uformer 256" "g2net 256" "fullsubnet 128"
bash tools/measure_round.sh r04 2>&1 | tail -12
