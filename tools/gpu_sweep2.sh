#!/bin/bash
# A/B env settings on the l training step: each argument is a quoted "VAR=v VAR2=v2" set
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
for v in "$@"; do
env $v timeout 600 python bench.py --workload train --model l --steps 10 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['roofline']['per_kind_ms']
print('$v', round(d['value'],1), round(d['ms_per_step'],3), {n:round(k[n],3) for n in k})" 2>&1 | tee -a gpurun_out/sweep2.log
done
