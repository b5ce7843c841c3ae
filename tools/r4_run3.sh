#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_uformer.py tests/test_gpu_dccrn.py -x -q -m gpu 2>&1 | tail -12
for g in 0 1 0 1; do echo "SE_UF_GAUSS=$g"; SE_UF_GAUSS=$g timeout 300 python tools/sweep.py --models uformer --batch 256 --steps 5 2>&1 | tail -1 | cut -c1-200; done
