#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for G in 2 1; do echo "gauss $G: $(SE_DCCRN_GAUSS=$G timeout 300 python tools/sweep.py --models dccrn --batch 256 --steps 5 2>&1 | grep utt_per_s | cut -c1-230)"; done
SE_DCCRN_GAUSS=2 timeout 900 python -m pytest tests/test_gpu_dccrn.py tests/test_gpu_full_fixture.py tests/test_gpu_b256_fixture.py tests/test_gpu_long_clips.py tests/test_gpu_ragged.py -x -q -m gpu -k "dccrn" 2>&1 | tail -3
