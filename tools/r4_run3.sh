#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for r in 1 2; do timeout 300 python tools/sweep.py --models uformer,taylorsenet,g2net,ctsnet,crn --batch 256 --steps 4 --no-profile 2>&1 | grep utt_per_s | cut -c1-80; done
