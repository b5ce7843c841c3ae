cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ts -o t -- python $GRAFT_REPO_ROOT/tools/stream_latency.py --models ctsnet_new --batch 1 --chunk 1 --seconds 1.0 > /tmp/ts.log 2>&1
python - <<'PY'
import csv,collections,glob
f=glob.glob('/tmp/ts/**/t_kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'rocclr' not in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# take the last 600 kernels (steady-state pushes)
tail=rows[-1500:]
g=collections.defaultdict(list)
for r in tail:
    n=r['Kernel_Name'].replace('void ','').replace('se::','').replace('(anonymous namespace)::','')[:40]
    g[n].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
busy=sum(sum(v) for v in g.values()); span=(int(tail[-1]['End_Timestamp'])-int(tail[0]['Start_Timestamp']))/1e3
print('kernels',len(tail),'busy us',round(busy),'span us',round(span))
for k,v in sorted(g.items(), key=lambda kv:-sum(kv[1]))[:14]:
    print(f"{k:40s} n={len(v):4d} avg {sum(v)/len(v):7.1f} us tot {sum(v):8.0f} us")
PY
