#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_uformer.py tests/test_gpu_long_clips.py -x -q -m gpu -k "uformer or Uformer" 2>&1 | tail -2
for r in 1 2; do timeout 300 python tools/sweep.py --models uformer --batch 256 --steps 5 --no-profile 2>&1 | grep utt_per_s | cut -c1-100; done
