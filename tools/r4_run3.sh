#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_dccrn.py tests/test_gpu_full_fixture.py tests/test_gpu_b256_fixture.py tests/test_gpu_long_clips.py tests/test_gpu_full_size.py tests/test_gpu_ragged.py tests/test_gpu_streaming.py tests/test_gpu_edge_cases.py tests/test_gpu_poison.py -x -q -m gpu -k "dccrn" 2>&1 | tail -15
for G in 1 0; do echo "gauss $G: $(SE_DCCRN_GAUSS=$G timeout 300 python tools/sweep.py --models dccrn --batch 256 --steps 5 2>&1 | grep utt_per_s | cut -c1-230)"; done
python - <<'PY'
import numpy as np, torch, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
import se_amd
from se_amd import synth
from se_amd.models import MODEL_CLASSES
outs={}
for g in ('1','0'):
    os.environ['SE_DCCRN_GAUSS']=g
    import subprocess
x = synth.synth_batch(4,'speech',64000,seed0=7)
PY
