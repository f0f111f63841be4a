#!/bin/bash
mkdir -p gpurun_out/f2
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/f2
(timeout 900 python -m pytest tests/test_kernels_conv.py -m gpu -q -x 2>&1 | tail -3) > $O/pytest_some.log 2>&1
tail -2 $O/pytest_some.log
(timeout 600 python tools/wgrad_probe.py --shapes 3,6,10,11,13,14,15 --variants 17/512,17/1024,33/1024,18/1024,49/256,49/512,49/1024,65/512,65/1024 2>&1 | tail -8) > $O/wgrad_probe.txt 2>&1
cat $O/wgrad_probe.txt
(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_l.json 2>&1
python -c "
import json,sys
d=json.load(open('$O/bench_train_l.json')); print(round(d['value'],1), round(d['ms_per_step'],2), d['config']['host_launch_ms_per_step'], d['roofline']['per_kind_ms'], round(d['roofline']['frac'],4))"
