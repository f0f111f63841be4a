#!/bin/bash
# A/B a tuning knob on the l training step: tools/gpu_sweep.sh VAR v1 v2 ...
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
VAR=$1; shift
for v in "$@"; do
env $VAR=$v timeout 600 python bench.py --workload train --model l --steps 10 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['roofline']['per_kind_ms']
print('$VAR=$v', round(d['value'],1), round(d['ms_per_step'],3), {n:k[n] for n in k if n.startswith('bn') or n.startswith('spp') or n=='view_copy'})" 2>&1 | tee -a gpurun_out/sweep_$VAR.log
done
