#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for d in 0 32 0 32; do
  echo "SE_GC_DBG=$d"
  SE_GC_DBG=$d timeout 900 python tools/sweep.py --models dccrn,fullsubnet,crn,uformer,gcrn --batch 256 --steps 5 --no-profile 2>&1 | grep utt_per_s | cut -c1-75
done
