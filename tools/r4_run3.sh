#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for lib in libse_prev.so libse_engine.so libse_prev.so libse_engine.so; do echo "== $lib"; SE_ENGINE_LIB=$ROOT/sixty-years-of-frequency-domain-monaural-speech-enhancement_amd/$lib timeout 600 python tools/sweep.py --models dccrn,uformer,g2net,dpcrn,ctsnet --batch 256 --steps 4 --no-profile 2>&1 | grep utt_per_s | cut -c1-75; SE_ENGINE_LIB=$ROOT/sixty-years-of-frequency-domain-monaural-speech-enhancement_amd/$lib timeout 600 python tools/sweep.py --models dccrn,g2net --batch 4 --steps 20 --no-profile 2>&1 | grep utt_per_s | cut -c1-75; done
