cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_b256_fixture.py tests/test_gpu_full_fixture.py tests/test_gpu_long_clips.py tests/test_gpu_ragged.py -q -m gpu -k "taylor" 2>&1 | grep -E "passed|failed" | cut -c1-150
for e in "SE_TAYLOR_FORK=0" "SE_TAYLOR_FORK=1"; do
 for m in taylorsenet taylorsenet_new; do
  (export $e; timeout 200 python tools/sweep.py --models $m --batch 256 --steps 5 --no-profile 2>&1 | grep utt_per_s | cut -c1-100 | sed "s/^/[$e] /")
 done
done
