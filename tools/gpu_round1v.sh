#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
for g in 0 1; do
(STREAMYOLO_AUTOTUNE=0 timeout 300 python bench.py --workload train --model l --batch 1 --height 64 --width 96 --steps 20 --warmup 5 --no-cpu-baseline --train-graph $g 2>&1 | tail -1) > gpurun_out/bench_tiny_$g.log 2>&1
python - gpurun_out/bench_tiny_$g.log <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("tiny l 64x96 B1 graph=%s" % d["config"]["hipgraph"], d["ms_per_step"], d["config"].get("host_launch_ms_per_step"), sum(d["roofline"]["per_kind_ms"].values()))
PY
done
