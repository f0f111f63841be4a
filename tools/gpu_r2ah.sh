#!/bin/bash
# round-2 stage ah: sibling convs merged in the inference plans (CSP conv2 + conv1, first cls + reg tower convs): tests + A/B
mkdir -p gpurun_out/ah
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/ah
(timeout 1200 python -m pytest tests/test_model_eval.py tests/test_streaming.py tests/test_optim.py tests/test_amp.py -m gpu -q -x 2>&1 | grep -vE 'RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids' | tail -3) > $O/pytest_eval.log 2>&1
tail -2 $O/pytest_eval.log
run() { tag=$1; shift; (env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $EXTRA 2>&1 | tail -1) > $O/bench_$tag.json 2>&1; }
EXTRA="--workload infer --model l" run infer_l_merged SY_X=1
EXTRA="--workload infer --model l" run infer_l_separate STREAMYOLO_MERGE_SIBLINGS=0
EXTRA="--workload infer --model s" run infer_s_merged SY_X=1
EXTRA="--workload infer --model s" run infer_s_separate STREAMYOLO_MERGE_SIBLINGS=0
EXTRA="--workload stream --model l --dtype fp16 --steps 50 --warmup 10 --u8-input 1" run stream_merged SY_X=1
EXTRA="--workload stream --model l --dtype fp16 --steps 50 --warmup 10 --u8-input 1" run stream_separate STREAMYOLO_MERGE_SIBLINGS=0
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],3), d.get('step_ms'))" 2>&1 | cut -c1-300; done
