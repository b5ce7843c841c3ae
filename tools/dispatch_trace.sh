#!/bin/bash
# per-dispatch kernel traces of tools/sweep.py (one model, 2 steps): gpurun_out/r4_trace/<model>_b<B>_trace.csv; bash tools/dispatch_trace.sh "uformer 256" "crn 64"
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  set -- $spec
  D=/tmp/tr_$1
  rm -rf $D
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python $ROOT/tools/sweep.py --models $1 --batch $2 --steps 2 --no-profile > $OUT/$1.log 2>&1
  f=$(find $D -name "*kernel_trace.csv" | head -1)
  # keep: kernel name (shortened), start, end, grid, workgroup
  python - "$f" "$OUT/$1_b$2_trace.csv" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
w=csv.writer(open(sys.argv[2],'w'))
w.writerow(['i','kernel','start_us','dur_us','grid','wg'])
for i,r in enumerate(rows):
    n=r['Kernel_Name']
    n=re.sub(r'^void ','',n); n=re.sub(r'\(anonymous namespace\)::','',n); n=re.sub(r'\(.*$','',n)
    w.writerow([i,n[:70],round((int(r['Start_Timestamp'])-t0)/1e3,1),round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,1),r.get('Grid_Size_X',r.get('Grid_Size','')),r.get('Workgroup_Size_X',r.get('Workgroup_Size',''))])
print(sys.argv[2],len(rows))
PY
  grep utt_per_s $OUT/$1.log | cut -c1-90
done
