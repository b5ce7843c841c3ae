#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_full_fixture.py tests/test_gpu_b256_fixture.py tests/test_gpu_long_clips.py tests/test_gpu_full_size.py tests/test_gpu_ragged.py tests/test_gpu_edge_cases.py -x -q -m gpu -k "lstm or crn or gcrn or fullsubnet" 2>&1 | tail -6
for B in 64 32 48; do timeout 600 python tools/sweep.py --models lstm,crn,gcrn --batch $B --steps 5 --no-profile 2>&1 | grep utt_per_s | cut -c1-100; done
for TC in 24 48 64; do echo "chunk $TC"; SE_LSTM_CHUNK_T=$TC timeout 600 python tools/sweep.py --models lstm,crn --batch 64 --steps 5 --no-profile 2>&1 | grep utt_per_s | cut -c1-100; done
echo "no chunk"; SE_LSTM_CHUNK=0 timeout 600 python tools/sweep.py --models lstm,crn --batch 64 --steps 5 --no-profile 2>&1 | grep utt_per_s | cut -c1-100
timeout 600 python tools/sweep.py --models lstm,crn,gcrn --batch 256 --steps 5 --no-profile 2>&1 | grep utt_per_s | cut -c1-100
