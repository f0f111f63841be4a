#!/bin/bash
# kernel timeline of the streaming step (graph replay): durations and gaps per kernel
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/trace_s
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_s -- python $GRAFT_REPO_ROOT/bench.py --workload stream --model l --dtype fp16 --steps 6 --warmup 4 --no-cpu-baseline $EXTRA 2>&1 | grep '^{"metric' | cut -c1-220
f=$(ls /tmp/trace_s/*/*kernel_trace.csv | head -1)
python - "$f" <<'P' | tee $GRAFT_REPO_ROOT/gpurun_out/trace_stream_summary.txt
import csv,sys,re
from collections import defaultdict
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]))
rows.sort()
# last step: from the last focus/frames kernel to the end
marks=[i for i,r in enumerate(rows) if "focus" in r[2] or "frames_to" in r[2]]
a,b=marks[-6],marks[-5]   # the last 3 focus launches belong to bench.py's per-op profile pass (no NMS)
step=rows[a:b]
t0=step[0][0]; t1=max(r[1] for r in step)
print("kernels in step: %d  wall %.3f ms  kernel sum %.3f ms"%(len(step),(t1-t0)/1e6,sum(e-s for s,e,_ in step)/1e6))
gaps=[step[i+1][0]-step[i][1] for i in range(len(step)-1)]
print("gaps: total %.3f ms, median %.2f us, max %.1f us"%(sum(gaps)/1e6,sorted(gaps)[len(gaps)//2]/1e3,max(gaps)/1e3))
fam=defaultdict(lambda:[0,0])
for s,e,n in step:
    k=re.sub(r"^void ","",n.replace("(anonymous namespace)::","").split("(")[0])[:60]
    fam[k][0]+=1; fam[k][1]+=e-s
for k,(n,d) in sorted(fam.items(),key=lambda kv:-kv[1][1])[:14]:
    print("%8.3f ms %4d  %s"%(d/1e6,n,k))
P
