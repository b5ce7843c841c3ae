#!/usr/bin/env python
"""Per-launch view of the tap-table GEMM family of one model: groups the `PROFL` lines the engine's profiler prints under
SE_PROF_DUMP=1 (one per timed launch: HIP-event ms, algorithmic GFLOP) by launch shape - where a model's family time sits and
at what TFLOP/s.

  SE_PROF_DUMP=1 python tools/sweep.py --models ctsnet --batch 256 --steps 2 2>&1 | python tools/profl.py
"""
import sys,collections
rows=[]
for l in sys.stdin:
    if l.startswith('PROFL'):
        p=l.split()
        rows.append((int(p[1]),float(p[2]),float(p[4]),float(p[6])))
# keep only the last pass (indices restart)
last=[]
for r in rows:
    if r[0]==0: last=[]
    last.append(r)
tot=sum(r[1] for r in last)
print('launches',len(last),'family ms',round(tot,2))
# group by (gflop rounded, ms bucket) identical launches
agg=collections.OrderedDict()
for i,ms,gf,tf in last:
    k=round(gf,1)
    a=agg.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=ms
for k,(n,ms) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:16]:
    print('%9.1f GFLOP x%3d  %7.3f ms each  %5.1f TF/s  %5.1f%% of family'%(k,n,ms/n,k/(ms/n) if ms else 0,100*ms/tot))
