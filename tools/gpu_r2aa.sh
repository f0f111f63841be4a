#!/bin/bash
# round-2 stage aa: co-scheduling knobs re-swept on the faster kernels (ring slots, wgrad workgroup cap, CSP fork, single stream)
mkdir -p gpurun_out/aa
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/aa
run() { tag=$1; shift; (env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_l_$tag.json 2>&1; }
run base SY_X=1
run ring3 STREAMYOLO_RING=3
run ring8 STREAMYOLO_RING=8
run cap256 STREAMYOLO_WGRAD_BLOCKS_CAP=256
run cap1024 STREAMYOLO_WGRAD_BLOCKS_CAP=1024
run nofork STREAMYOLO_CSP_FORK=0
run streams1 STREAMYOLO_STREAMS=1
run base2 SY_X=1
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],3), d.get('step_ms'))" 2>&1 | cut -c1-300; done
