#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_uformer.py tests/test_gpu_new_variants.py -x -q -m gpu 2>&1 | tail -2
for k in 128 0 128 0; do echo "SE_GC_PW_BM64=$k"; SE_GC_PW_BM64=$k timeout 600 python tools/sweep.py --models uformer,dpcrn,gcrn,taylorsenet,crn,dccrn --batch 256 --steps 4 --no-profile 2>&1 | grep utt_per_s | cut -c1-75; done
