#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_uformer.py -x -q -m gpu 2>&1 | tail -3
for f in 0 1; do echo "SE_UF_FOLD=$f"; SE_UF_FOLD=$f timeout 300 python tools/sweep.py --models uformer --batch 256 --steps 5 2>&1 | tail -1 | cut -c1-200; done
timeout 300 python tools/sweep.py --models dccrn,g2net,crn --batch 256 --steps 5 2>&1 | tail -3 | cut -c1-120
