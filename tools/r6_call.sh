#!/bin/bash
# round-6 GPU calls (through gpurun): bash tools/r6_call.sh <step> ...
#   steps: tests[:k-expr] parity measure pmc:<model>:<B> prof:<model>:<B> profl:<model>:<B>[:env=val,..] dbg:<model>:<B>:<env=val,..>
#          fsnsweep:<lo>:<hi> step:<H>:<S>[:env=val,..] stept:<H>:<S> bench[:steps] corpus stream sweep:<B>
#          slat:<models>[:env=val,..] (push latencies) sprof:<model>[:env=val,..] (kernel summary of 200 one-frame pushes)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PKG=$(ls -d $ROOT/*_amd)
for step in "$@"; do
  IFS=: read -r kind a b c <<< "$step"
  case $kind in
    tests)
      (cd $ROOT && timeout 1700 python -m pytest tests -x -q -m gpu ${a:+-k "$a"} 2>&1 | tail -15) > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log ;;
    parity)
      (cd $ROOT && timeout 900 python tools/parity_record.py --out $OUT/r06_parity.json > $OUT/parity.log 2>&1; tail -2 $OUT/parity.log) ;;
    measure)
      bash $ROOT/tools/measure_round.sh r06 ;;
    pmc)
      bash $ROOT/tools/pmc_models.sh r06 "$a $b" ;;
    prof)
      D=$OUT/prof_${a}_b${b}
      timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o s -- python $ROOT/tools/sweep.py --models $a --batch $b --steps 3 --no-profile --fsn-max-batch 256 > $D.log 2>&1
      cp $D/s_kernel_stats.csv $OUT/r06_${a}_b${b}_kernel_stats.csv 2>/dev/null
      grep utt_per_s $D.log | cut -c1-120
      head -8 $OUT/r06_${a}_b${b}_kernel_stats.csv | cut -c1-150
      rm -rf $D ;;
    profl)
      (export SE_PROF_DUMP=1 SE_R6=1 ${c:+${c//,/ }}; timeout 300 python $ROOT/tools/sweep.py --models $a --batch $b --steps 2 --fsn-max-batch 256 2>&1 | python $ROOT/tools/profl.py > $OUT/profl_${a}_b${b}${c:+_${c//[=,]/_}}.txt; head -14 $OUT/profl_${a}_b${b}${c:+_${c//[=,]/_}}.txt) ;;
    dbg)
      (export SE_R6=1 ${c:+${c//,/ }}; timeout 300 python $ROOT/tools/sweep.py --models $a --batch $b --steps 5 --no-profile --fsn-max-batch 256 2>&1 | grep utt_per_s | cut -c1-100 | sed "s/^/[$c] /") ;;
    len)   # len:<model>:<B>:<samples>[,env=val..]
      (IFS=, read -r smp envs <<< "$c"; export SE_R6=1 ${envs:+${envs//;/ }}; timeout 300 python $ROOT/tools/sweep.py --models $a --batch $b --steps 5 --no-profile --samples $smp 2>&1 | grep utt_per_s | cut -c1-100 | sed "s/^/[L=$smp $envs] /") ;;
    rag)   # rag:<model>:<B>[:env=val,..]: the ragged pass of tools/sweep.py (512 clips of 512 lengths through the batch plan)
      (export SE_R6=1 ${c:+${c//,/ }}; timeout 400 python $ROOT/tools/sweep.py --models $a --batch $b --steps 2 --no-profile --ragged 512 2>&1 | grep utt_per_s | python -c "import sys,json; [print('[${c:-}]', d['model'], 'ragged utt/s', d.get('ragged_utt_per_s'), 'x rt', d.get('ragged_x_realtime')) for d in map(json.loads, sys.stdin)]") ;;
    fsnsweep)
      : > $OUT/r06_fsn_batch_sweep.jsonl
      for B in $(seq $a $b); do
        timeout 200 python $ROOT/tools/sweep.py --models fullsubnet --batch $B --steps 4 --no-profile --fsn-max-batch 256 2>&1 | grep utt_per_s >> $OUT/r06_fsn_batch_sweep.jsonl
      done
      cut -c1-90 $OUT/r06_fsn_batch_sweep.jsonl ;;
    step)
      (export SE_R6=1 ${c:+${c//,/ }}; $PKG/gcbench step $a $b 2>&1 | tail -1 | sed "s/^/[${c:-}] /") ;;
    gcb)   # gcb:<"Cin Cout Fin B T">:<env=val,..>
      (export SE_R6=1 ${b:+${b//,/ }}; $PKG/gcbench $a 2>&1 | tail -2 | tr '\n' ' ' | sed "s/^/[${b:-}] /"; echo) ;;
    gcbp)   # the previous build's gcbench (same-box A/B of a kernel change)
      (export SE_R6=1 ${b:+${b//,/ }}; $PKG/gcbench_prev $a 2>&1 | tail -2 | tr '\n' ' ' | sed "s/^/[prev ${b:-}] /"; echo) ;;
    dbgp)   # dbgp:<model>:<B>[:env]: the previous build of the library
      (export SE_ENGINE_LIB=$PKG/libse_engine_prev.so ${c:+${c//,/ }}; timeout 300 python $ROOT/tools/sweep.py --models $a --batch $b --steps 5 --no-profile --fsn-max-batch 256 2>&1 | grep utt_per_s | cut -c1-100 | sed "s/^/[prev ${c:-}] /") ;;
    gcbt)   # gcbt:<"Cin Cout Fin B T">:<env=val,..>  (phase-timing build)
      (export SE_R6=1 ${b:+${b//,/ }}; $PKG/gcbench_timing $a 2>&1 | tail -4 | cut -c1-400 | sed "s/^/[${b:-}] /") ;;
    stept)
      $PKG/gcbench_timing step $a $b 2>&1 | tail -2 ;;
    bench)
      (cd $ROOT && timeout 900 python bench.py --steps ${a:-10} --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; python tools/bench_digest.py $OUT/bench.json) ;;
    benchz)   # benchz[:env=val,..]: bench.py (3 steps), zoo rows only in the digest
      (cd $ROOT && export SE_R6=1 ${a:+${a//,/ }}; timeout 900 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/benchz.json 2> $OUT/benchz.err; python tools/bench_digest.py $OUT/benchz.json | grep -E "taylor|g2net|dccrn" | sed "s/^/[${a:-}] /") ;;
    corpus)
      (cd $ROOT && timeout 600 python tools/corpus_bench.py > $OUT/r06_corpus.json 2> $OUT/corpus.err; tail -c 600 $OUT/r06_corpus.json) ;;
    stream)
      (cd $ROOT && timeout 600 python tools/stream_latency.py > $OUT/r06_stream_latency.jsonl 2> $OUT/stream.err; cut -c1-140 $OUT/r06_stream_latency.jsonl | head -30) ;;
    sweep)
      (cd $ROOT && timeout 900 python tools/sweep.py --batch $a --models lstm,crn,gcrn,dpcrn,dccrn,fullsubnet,ctsnet,g2net,taylorsenet,uformer,ctsnet_new,g2net_new,taylorsenet_new > $OUT/r06_sweep_b$a.jsonl 2> $OUT/sweep.err; cut -c1-110 $OUT/r06_sweep_b$a.jsonl) ;;
    slat)   # slat:<models>[:env=val,..]: one-frame / eight-frame push latencies, 1 and 16 streams
      (cd $ROOT && export SE_R6=1 ${b:+${b//,/ }}; timeout 400 python tools/stream_latency.py --models $a 2>&1 | grep ms_per_push | python -c "import sys,json; [print(\"[${b:-}]\", d[\"model\"], \"streams\", d[\"streams\"], \"frames\", d[\"frames_per_push\"], \"ms\", d[\"ms_per_push\"], \"p95\", d[\"p95_ms\"], \"host enqueue\", d.get(\"host_enqueue_ms\")) for d in map(json.loads, sys.stdin)]") ;;
    sprof)   # sprof:<model>[:env=val,..]: kernel summary of one-frame pushes (1 stream, 1 s of audio, 2 passes = 200 pushes)
      D=$OUT/sprof_$a
      (export SE_R6=1 ${b:+${b//,/ }}; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o s -- python $ROOT/tools/stream_latency.py --models $a --batch 1 --chunk 1 --seconds 1 > $D.log 2>&1)
      grep ms_per_push $D.log | cut -c1-120
      python - $D/s_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
calls = sum(int(r['Calls']) for r in rows); tot = sum(float(r['TotalDurationNs']) for r in rows)
print('launches / push ~', round(calls / 200.0, 1), ' GPU us / push ~', round(tot / 200.0 / 1e3, 1))
for r in rows[:14]:
    print('%6d %9.1f us/push  avg %7.2f us  %s' % (int(r['Calls']), float(r['TotalDurationNs']) / 200e3, float(r['AverageNs']) / 1e3, r['Name'][:90]))
PY
      cp $D/s_kernel_stats.csv $OUT/r06_push_${a}_kernel_stats.csv 2>/dev/null
      rm -rf $D ;;
    *) echo "unknown step $step" ;;
  esac
done
