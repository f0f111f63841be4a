#!/bin/bash
# round-2 stage t: BatchNorm finalize fused into the apply launch: tests + A/B on one box
mkdir -p gpurun_out/t
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/t
(timeout 900 python -m pytest tests/test_kernels_misc.py tests/test_model_train.py -m gpu -q -x 2>&1 | grep -vE 'RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids' | tail -4) > $O/pytest_train.log 2>&1
tail -2 $O/pytest_train.log
run() { tag=$1; shift; (env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_l_$tag.json 2>&1; }
run fused SY_X=1
run separate STREAMYOLO_FUSED_FINALIZE=0
run fused_copies8 STREAMYOLO_STAT_COPIES=8
run fused_cap1024 SY_BN_FAPPLY_BLOCKS=1024
run fused_cap4096 SY_BN_FAPPLY_BLOCKS=4096
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],3), d.get('step_ms'), d['roofline'].get('per_kind_ms'))" 2>&1 | cut -c1-500; done
