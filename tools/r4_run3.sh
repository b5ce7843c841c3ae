#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_b256_fixture.py tests/test_gpu_streaming.py -x -q -m gpu 2>&1 | tail -2
for r in 1 2; do timeout 300 python tools/sweep.py --models crn,gcrn,lstm --batch 256 --steps 5 --no-profile 2>&1 | grep utt_per_s | cut -c1-80; done
timeout 300 python tools/sweep.py --models crn --batch 64 --steps 10 --no-profile 2>&1 | grep utt_per_s | cut -c1-80
