#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_dccrn.py tests/test_gpu_models.py tests/test_gpu_new_variants.py -x -q -m gpu 2>&1 | tail -2
for r in 1 2; do timeout 900 python tools/sweep.py --models dccrn,uformer,g2net,dpcrn,crn,gcrn,ctsnet --batch 256 --steps 4 --no-profile 2>&1 | grep utt_per_s | cut -c1-75; done
timeout 300 python tools/sweep.py --models crn,dccrn,gcrn --batch 1 --steps 40 --no-profile 2>&1 | grep utt_per_s | cut -c1-75
timeout 300 python tools/sweep.py --models dccrn,g2net --batch 8 --steps 10 --no-profile 2>&1 | grep utt_per_s | cut -c1-75
