#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "fullsubnet" 2>&1 | tail -2
for s in 1 2 3 4 6; do echo "SPLIT=$s"; SE_FSN_SPLIT=$s timeout 300 python tools/sweep.py --models fullsubnet --batch 128 --steps 3 --no-profile 2>&1 | tail -1 | cut -c1-100; done
SE_GC_WIDE128=0 timeout 300 python tools/sweep.py --models fullsubnet --batch 128 --steps 3 --no-profile 2>&1 | tail -1 | cut -c1-100
