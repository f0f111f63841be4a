#!/bin/bash
# round-2 stage ak: new BatchNorm grid caps (defaults) vs the old ones on l / s / m, GPU tests
mkdir -p gpurun_out/ak
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/ak
(timeout 900 python -m pytest tests/test_kernels_misc.py tests/test_model_train.py -m gpu -q -x 2>&1 | grep -vE 'RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids' | tail -2) > $O/pytest.log 2>&1
tail -1 $O/pytest.log
run() { tag=$1; shift; (env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $EXTRA 2>&1 | tail -1) > $O/bench_$tag.json 2>&1; }
OLD="SY_BN_APPLY_BLOCKS=4096 SY_BN_REDUCE_BLOCKS=1024 SY_BN_BAPPLY_BLOCKS=2048"
run l_new SY_X=1
run l_old $OLD
EXTRA="--model s" run s_new SY_X=1
EXTRA="--model s" run s_old $OLD
EXTRA="--model m" run m_new SY_X=1
EXTRA="--model m" run m_old $OLD
EXTRA="--batch 4" run l_b4_new SY_X=1
EXTRA="--batch 4" run l_b4_old $OLD
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],3))" 2>&1 | cut -c1-300; done
