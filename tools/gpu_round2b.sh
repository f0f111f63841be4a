#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
for cap in 4096 2048 1024 512; do
echo "== cap $cap"
SY_BN_APPLY_BLOCKS=$cap SY_BN_REDUCE_BLOCKS=$cap SY_BN_BAPPLY_BLOCKS=$cap timeout 600 python bench.py --workload train --model l --steps 6 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['per_kind_ms']
print(d['ms_per_step'], {n:k[n] for n in k if n.startswith('bn')})"
done > gpurun_out/bn_caps.txt 2>&1
cat gpurun_out/bn_caps.txt
