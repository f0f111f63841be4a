#!/bin/bash
# round-2 stage l: channel-sliced BatchNorm backward reduce (fewer closing atomics): tests + A/B bench on one box
mkdir -p gpurun_out/l
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/l
(timeout 900 python -m pytest tests/test_kernels_misc.py tests/test_model_train.py -m gpu -q -x -k "bn or segmented or train_step_s or tape" 2>&1 | grep -vE 'RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids' | tail -4) > $O/pytest_bn.log 2>&1
tail -2 $O/pytest_bn.log
run() { tag=$1; shift; (env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_l_$tag.json 2>&1; }
run sliced64 SY_X=1
run unsliced SY_BN_REDUCE_SLICE=4096
run sliced64_cap512 SY_BN_REDUCE_BLOCKS=512
run sliced64_cap2048 SY_BN_REDUCE_BLOCKS=2048
run sliced32 SY_BN_REDUCE_SLICE=32
run sliced128 SY_BN_REDUCE_SLICE=128
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],3), d.get('step_ms'), round(d['roofline']['frac'],4), d['roofline'].get('per_kind_ms'))" 2>&1 | cut -c1-600; done
