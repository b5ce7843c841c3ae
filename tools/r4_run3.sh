#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for d in 0 16 0 16; do echo "SE_GC_DBG=$d"; SE_GC_DBG=$d timeout 300 python tools/sweep.py --models dccrn --batch 256 --steps 6 2>&1 | tail -1 | cut -c1-200; done
