# kernel summaries of tools/sweep.py per model: bash tools/prof_models.sh "crn 64" "uformer 256" ...
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  set -- $spec
  OUT=$ROOT/gpurun_out/r3_prof_$1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python $ROOT/tools/sweep.py --models $1 --batch $2 --steps 3 --no-profile > $OUT.log 2>&1
  cp $OUT/s_kernel_stats.csv $ROOT/gpurun_out/r03_$1_b$2_kernel_stats.csv 2>/dev/null
  grep utt_per_s $OUT.log | cut -c1-90
done
