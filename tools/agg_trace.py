#!/usr/bin/env python
"""Per (kernel, grid) totals of a rocprofv3 --kernel-trace csv: which launch shapes of a kernel the time sits in.

  python tools/agg_trace.py <..._kernel_trace.csv>   ->  name, LDS, grid, VGPRs, launches, mean us, total us (gc_kernel / tcm only)
"""
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.OrderedDict()
for r in rows:
    n=r['Kernel_Name']
    if 'gc_kernel' not in n and 'tcm' not in n: continue
    k=(n[:70],r.get('LDS_Block_Size'),r.get('Grid_Size_X', r.get('Grid_Size')),r.get('VGPR_Count'),r.get('Accum_VGPR_Count'))
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    a=agg.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=d
for k,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:40]:
    print('%-72s lds %6s grid %9s vgpr %4s/%4s  n %4d  avg %8.1f us tot %8.1f'%(k[0],k[1],k[2],k[3],k[4],c,t/c,t))
