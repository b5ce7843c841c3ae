"""Summarise a rocprofv3 --kernel-trace --stats --output-format csv run under /tmp/st: top kernels (calls, avg us, share) and
the busy / span time of the trace (tuning helper for the frame-online mode)."""
import csv,glob,sys
f=glob.glob("/tmp/st/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-62s %6s %9.1f %6s" % (r["Name"][:62], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
f=glob.glob("/tmp/st/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
n=len(rows); tot=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows)
print("kernels",n,"busy ms",tot/1e6,"span ms",(int(rows[-1]["End_Timestamp"])-int(rows[0]["Start_Timestamp"]))/1e6)
