#!/bin/bash
# Round measurements on the GPU box (run through gpurun): the driver's bench line, the rocprofv3 kernel summary of the same
# command, and the three separate --pmc passes tools/pmc_summary.py reads.  Every step has its own timeout.
#   bash tools/measure_round.sh r02      -> gpurun_out/r02_m/
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${TAG}_m
mkdir -p $OUT $OUT/pmc
cd /tmp && export TMPDIR=/tmp
timeout 400 python $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o s -- \
    python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-configs > $OUT/kt.log 2>&1
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32"; do
    D=$OUT/pmc/$(echo $C | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o p -- \
        python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-profile > $D.log 2>&1
done
# the summaries are made here (the raw counter files are tens of MB each: gpurun merges back at most 64 MiB)
python $ROOT/tools/pmc_summary.py $OUT/pmc $OUT/${TAG}_pmc_dccrn --steps 3 | cut -c1-200
cp $OUT/kt/s_kernel_stats.csv $OUT/${TAG}_dccrn_b256_kernel_stats.csv 2>/dev/null
rm -rf $OUT/pmc/*/p_*.csv $OUT/kt/s_kernel_trace.csv
ls -la $OUT | head -20
tail -c 600 $OUT/bench.json
