# kernel trace of steady-state one-frame pushes: kernels per push, busy vs span, top kernels.  usage: trace_stream.sh <model> [chunk] [streams]
M=${1:-ctsnet_new}; C=${2:-1}; NB=${3:-1}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ts
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ts -o t -- python $GRAFT_REPO_ROOT/tools/stream_latency.py --models $M --batch $NB --chunk $C --seconds 1.0 > /tmp/ts.log 2>&1
python - "$M" <<'PY'
import csv,collections,glob,sys
f=glob.glob('/tmp/ts/**/t_kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
tail=rows[-3000:]
# pushes are delimited by the STFT kernel
def nm(r): return r['Kernel_Name'].replace('void ','').replace('se::','').replace('(anonymous namespace)::','')[:44]
starts=[i for i,r in enumerate(tail) if 'stft' in nm(r) and 'istft' not in nm(r)]
per=[starts[i+1]-starts[i] for i in range(len(starts)-1)]
g=collections.defaultdict(list)
a,b=starts[0],starts[-1]
seg=tail[a:b]
for r in seg: g[nm(r)].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
busy=sum(sum(v) for v in g.values()); span=(int(tail[b]['Start_Timestamp'])-int(tail[a]['Start_Timestamp']))/1e3
np_=len(starts)-1
print(sys.argv[1],'pushes',np_,'kernels/push',len(seg)/np_,'busy us/push',round(busy/np_),'span us/push',round(span/np_))
for k,v in sorted(g.items(), key=lambda kv:-sum(kv[1]))[:16]:
    print(f"{k:44s} n/push={len(v)/np_:5.1f} avg {sum(v)/len(v):7.1f} us tot/push {sum(v)/np_:7.0f} us")
PY
