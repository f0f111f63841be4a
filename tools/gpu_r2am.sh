#!/bin/bash
# round-2 stage am: 16 statistic replicas with / without the fused finalize+apply launch: reproducibility tests + bench
mkdir -p gpurun_out/am
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/am
(STREAMYOLO_STAT_COPIES=16 STREAMYOLO_FUSED_FINALIZE=1 timeout 1200 python -m pytest tests/test_model_train.py tests/test_amp.py tests/test_optim.py -m gpu -q 2>&1 | grep -vE 'RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids' | tail -4) > $O/pytest_copies16_fused.log 2>&1
tail -3 $O/pytest_copies16_fused.log
run() { tag=$1; shift; (env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_l_$tag.json 2>&1; }
run base SY_X=1
run copies16 STREAMYOLO_STAT_COPIES=16
run copies16_fused STREAMYOLO_STAT_COPIES=16 STREAMYOLO_FUSED_FINALIZE=1
run copies32_fused STREAMYOLO_FUSED_FINALIZE=1
run base2 SY_X=1
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],3))" 2>&1 | cut -c1-300; done
