#!/bin/bash
# round-2 stage ag: halo2 pixel-fragment read-ahead 3 steps (BD = 4) vs 2 (previous build)
mkdir -p gpurun_out/ag
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/ag
PREV=$GRAFT_REPO_ROOT/streamyolo_amd/lib/ab_prev.so
for m in stats dgrad; do
echo "== new (BD 4), $m" >> $O/halo2_bd.txt
(timeout 300 python tools/conv_probe.py --mode $m --shapes 3,6,10,11,13,14 --tiles 117,118 --reps 9 --chain 4 2>&1 | grep -v amdgpu.ids) >> $O/halo2_bd.txt 2>&1
echo "== previous build (BD 3), $m" >> $O/halo2_bd.txt
(STREAMYOLO_HIP_LIB=$PREV timeout 300 python tools/conv_probe.py --mode $m --shapes 3,6,10,11,13,14 --tiles 117,118 --reps 9 --chain 4 2>&1 | grep -v amdgpu.ids) >> $O/halo2_bd.txt 2>&1
done
cat $O/halo2_bd.txt
run() { tag=$1; shift; (env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_l_$tag.json 2>&1; }
run new SY_X=1
run prev STREAMYOLO_HIP_LIB=$PREV
run new2 SY_X=1
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],3), {k: v for k, v in d['roofline'].get('per_kind_ms').items() if k in ('conv','dgrad','wgrad')})" 2>&1 | cut -c1-300; done
