set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for spec in "crn 64" "uformer 256" "fullsubnet 128" "lstm 1"; do
  set -- $spec
  OUT=$ROOT/gpurun_out/r3_prof_$1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python $ROOT/tools/sweep.py --models $1 --batch $2 --steps 3 --no-profile > $OUT.log 2>&1
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  cp $f $ROOT/gpurun_out/r3_$1_b$2_kernel_stats.csv
  tail -1 $OUT.log
done
