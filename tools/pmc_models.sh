#!/bin/bash
# PMC passes (separate runs per counter set, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) of tools/sweep.py for the
# models named: bash tools/pmc_models.sh r04 "crn 64" "uformer 256" ... -> gpurun_out/<tag>_pmc_<model>_b<B>.{json,md}
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  set -- $spec
  OUT=$ROOT/gpurun_out/${TAG}_pmcraw_$1
  mkdir -p $OUT
  for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32"; do
    D=$OUT/$(echo $C | cut -d' ' -f1)
    timeout 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o p -- \
        python $ROOT/tools/sweep.py --models $1 --batch $2 --steps 1 --no-profile > $D.log 2>&1
  done
  python $ROOT/tools/pmc_summary.py $OUT $ROOT/gpurun_out/${TAG}_pmc_$1_b$2 --steps 3 | cut -c1-200
  rm -rf $OUT/*/p_kernel_trace.csv $OUT/*/p_counter_collection.csv
  du -sh $OUT | cut -c1-60
done
