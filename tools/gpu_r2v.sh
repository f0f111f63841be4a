#!/bin/bash
# round-2 stage v: leaner wgrad9 (constant waits, branch-free DMA offsets, pipelined fragment reads) and M0-constrained LDS-DMA in every kernel:
# kernel tests, A/B probes and bench against the previous library build (streamyolo_amd/lib/ab_prev.so) on one box
mkdir -p gpurun_out/v
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/v
PREV=$GRAFT_REPO_ROOT/streamyolo_amd/lib/ab_prev.so
(timeout 900 python -m pytest tests/test_kernels_conv.py tests/test_kernels_misc.py -m gpu -q -x 2>&1 | grep -vE 'RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids' | tail -3) > $O/pytest_kernels.log 2>&1
tail -2 $O/pytest_kernels.log
echo "== new" > $O/wgrad9_ab.txt
(timeout 300 python tools/wgrad_probe.py --shapes 3,6,10,11,13,14 --variants 65/256,49/256,67/256,65/512 --reps 7 --chain 3 2>&1 | grep -v amdgpu.ids) >> $O/wgrad9_ab.txt 2>&1
echo "== previous build" >> $O/wgrad9_ab.txt
(STREAMYOLO_HIP_LIB=$PREV timeout 300 python tools/wgrad_probe.py --shapes 3,6,10,11,13,14 --variants 65/256,49/256,67/256,65/512 --reps 7 --chain 3 2>&1 | grep -v amdgpu.ids) >> $O/wgrad9_ab.txt 2>&1
cat $O/wgrad9_ab.txt
echo "== new" > $O/conv_ab.txt
(timeout 300 python tools/conv_probe.py --mode stats --shapes 10,6,13,9,5 --tiles 117,118,121,86 --reps 7 --chain 4 2>&1 | grep -v amdgpu.ids) >> $O/conv_ab.txt 2>&1
echo "== previous build" >> $O/conv_ab.txt
(STREAMYOLO_HIP_LIB=$PREV timeout 300 python tools/conv_probe.py --mode stats --shapes 10,6,13,9,5 --tiles 117,118,121,86 --reps 7 --chain 4 2>&1 | grep -v amdgpu.ids) >> $O/conv_ab.txt 2>&1
cat $O/conv_ab.txt
run() { tag=$1; shift; (env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_l_$tag.json 2>&1; }
run new SY_X=1
run prev STREAMYOLO_HIP_LIB=$PREV
run new2 SY_X=1
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],3), d.get('step_ms'), d['roofline'].get('per_kind_ms'))" 2>&1 | cut -c1-500; done
