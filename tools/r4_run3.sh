#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for s in 1 2 3 4; do echo "SPLIT=$s"; SE_FSN_SPLIT=$s timeout 300 python tools/sweep.py --models fullsubnet --batch 128 --steps 3 --no-profile 2>&1 | tail -1 | cut -c1-110; done
for b in 96 112 120 127; do timeout 300 python tools/sweep.py --models fullsubnet --batch $b --steps 3 --no-profile 2>&1 | tail -1 | cut -c1-110; done
