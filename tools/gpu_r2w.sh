#!/bin/bash
# round-2 stage w: leaner transpose-read wgrad (1x1 / stride-2 layers; linear 1x1 addressing) vs the previous build on one box
mkdir -p gpurun_out/w
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/w
PREV=$GRAFT_REPO_ROOT/streamyolo_amd/lib/ab_prev.so
(timeout 900 python -m pytest tests/test_kernels_conv.py -m gpu -q -x -k wgrad 2>&1 | tail -2) > $O/pytest_wgrad.log 2>&1
tail -1 $O/pytest_wgrad.log
V=18/256,18/512,17/256,17/512,33/512,34/512,20/512,21/512
echo "== new" > $O/wgrad_tr_ab.txt
(timeout 300 python tools/wgrad_probe.py --shapes 9,5,7,16,2,1,8 --variants $V --reps 7 --chain 3 2>&1 | grep -v amdgpu.ids) >> $O/wgrad_tr_ab.txt 2>&1
echo "== previous build" >> $O/wgrad_tr_ab.txt
(STREAMYOLO_HIP_LIB=$PREV timeout 300 python tools/wgrad_probe.py --shapes 9,5,7,16,2,1,8 --variants $V --reps 7 --chain 3 2>&1 | grep -v amdgpu.ids) >> $O/wgrad_tr_ab.txt 2>&1
cat $O/wgrad_tr_ab.txt
run() { tag=$1; shift; (env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_l_$tag.json 2>&1; }
run new SY_X=1
run prev STREAMYOLO_HIP_LIB=$PREV
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],3), d.get('step_ms'), d['roofline'].get('per_kind_ms'))" 2>&1 | cut -c1-500; done
