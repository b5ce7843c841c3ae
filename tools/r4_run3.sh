#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "fullsubnet" -s 2>&1 | tail -15
