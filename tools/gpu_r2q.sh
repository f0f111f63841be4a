#!/bin/bash
# round-2 stage q: wgrad9 on eight waves (51 / 67) vs four (49 / 65), with load ablations
mkdir -p gpurun_out/q
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/q
(timeout 600 python -m pytest tests/test_kernels_conv.py -m gpu -q -x -k "all_taps" 2>&1 | tail -2) > $O/pytest_wgrad9.log 2>&1
tail -1 $O/pytest_wgrad9.log
(timeout 300 python tools/wgrad_probe.py --shapes 3,6,10,11,13,14 --variants 65/256,67/256,67/128,67/512,51/256,835/256 --reps 7 --chain 3 2>&1 | grep -v amdgpu.ids) > $O/wgrad9b.txt 2>&1
cat $O/wgrad9b.txt
(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_l.json 2>&1
python -c "
import json
d=json.load(open('$O/bench_train_l.json')); print(round(d['value'],1), round(d['ms_per_step'],3), d.get('step_ms'), d['roofline'].get('per_kind_ms'))"
(timeout 600 python tools/profile_train.py 2>&1 | grep -v amdgpu.ids | grep "wgrad") > $O/profile_wgrad.txt 2>&1
head -40 $O/profile_wgrad.txt
