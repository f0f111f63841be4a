#!/bin/bash
# round-2 stage y: hipGraph replay vs launch tapes for the training step (batch 8 / 4, model s)
mkdir -p gpurun_out/y
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/y
run() { tag=$1; shift; (timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>&1 | tail -1) > $O/bench_$tag.json 2>&1; }
run l_b8_tape
run l_b8_graph --train-graph 1
run l_b4_tape --batch 4
run l_b4_graph --batch 4 --train-graph 1
run s_b8_tape --model s
run s_b8_graph --model s --train-graph 1
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],3), d.get('step_ms'), d['config'].get('host_launch_ms_per_step'))" 2>&1 | cut -c1-300; done
