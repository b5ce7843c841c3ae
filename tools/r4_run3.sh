#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "fullsubnet or lstm" 2>&1 | tail -2
for lib in libse_prev.so libse_engine.so libse_prev.so libse_engine.so; do echo "== $lib $(SE_ENGINE_LIB=$ROOT/sixty-years-of-frequency-domain-monaural-speech-enhancement_amd/$lib timeout 300 python tools/sweep.py --models fullsubnet --batch 128 --steps 4 --no-profile 2>&1 | grep utt_per_s | cut -c28-90)"; done
