#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for lib in libse_4bfc897.so libse_engine.so libse_4bfc897.so libse_engine.so; do echo "== $lib"; SE_ENGINE_LIB=$ROOT/sixty-years-of-frequency-domain-monaural-speech-enhancement_amd/$lib timeout 300 python tools/sweep.py --models crn,dccrn,gcrn --batch 1 --steps 40 --no-profile 2>&1 | grep utt_per_s | cut -c1-75; done
for b in 8 32; do for lib in libse_4bfc897.so libse_engine.so; do echo "B=$b $lib $(SE_ENGINE_LIB=$ROOT/sixty-years-of-frequency-domain-monaural-speech-enhancement_amd/$lib timeout 300 python tools/sweep.py --models dccrn,g2net --batch $b --steps 10 --no-profile 2>&1 | grep utt_per_s | cut -c28-60 | tr '\n' ' ')"; done; done
for r in 1 2; do timeout 900 python tools/sweep.py --models dccrn,fullsubnet,uformer,g2net,dpcrn,crn --batch 256 --steps 4 --no-profile 2>&1 | grep utt_per_s | cut -c1-75; done
