#!/bin/bash
# round-5 GPU calls (through gpurun): bash tools/r5_call.sh <step> ...   steps: tests parity prof:<model>:<B> dbg:<model>:<B>:<env=val,..> bench
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PKG=$(ls -d $ROOT/*_amd)
for step in "$@"; do
  IFS=: read -r kind a b c <<< "$step"
  case $kind in
    tests)
      (cd $ROOT && timeout 1500 python -m pytest tests -x -q -m gpu ${a:+-k "$a"} 2>&1 | tail -15) > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log ;;
    parity)
      (cd $ROOT && timeout 900 python tools/parity_record.py --out $OUT/r05_parity.json > $OUT/parity.log 2>&1; tail -2 $OUT/parity.log
       SE_ENGINE_LIB=$PKG/libse_engine_exact.so timeout 900 python tools/parity_record.py --out $OUT/r05_parity_exact.json > $OUT/parity_exact.log 2>&1; tail -2 $OUT/parity_exact.log) ;;
    parityx)
      (cd $ROOT && SE_ENGINE_LIB=$PKG/libse_engine_exact.so timeout 900 python tools/parity_record.py --out $OUT/r05_parity_exact.json > $OUT/parity_exact.log 2>&1; tail -1 $OUT/parity_exact.log) ;;
    measure)
      bash $ROOT/tools/measure_round.sh r05 ;;
    pmc)
      bash $ROOT/tools/pmc_models.sh r05 "$a $b" ;;
    prof)
      D=$OUT/prof_${a}_b${b}
      timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o s -- python $ROOT/tools/sweep.py --models $a --batch $b --steps 3 --no-profile > $D.log 2>&1
      cp $D/s_kernel_stats.csv $OUT/r05_${a}_b${b}_kernel_stats.csv 2>/dev/null
      grep utt_per_s $D.log | cut -c1-120
      head -8 $OUT/r05_${a}_b${b}_kernel_stats.csv | cut -c1-150
      rm -rf $D ;;
    dbg)
      (export ${c//,/ }; timeout 200 python $ROOT/tools/sweep.py --models $a --batch $b --steps 5 --no-profile 2>&1 | grep utt_per_s | cut -c1-100 | sed "s/^/[$c] /") ;;
    bench)
      (cd $ROOT && timeout 600 python bench.py --steps ${a:-10} --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; python tools/bench_digest.py $OUT/bench.json) ;;
    *) echo "unknown step $step" ;;
  esac
done
