#!/bin/bash
# round-2 stage ac: residual adds on the staged (lean) conv epilogue: eval tests + inference / streaming A/B vs the previous build
mkdir -p gpurun_out/ac
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/ac
PREV=$GRAFT_REPO_ROOT/streamyolo_amd/lib/ab_prev.so
(timeout 1200 python -m pytest tests/test_kernels_conv.py tests/test_model_eval.py tests/test_streaming.py -m gpu -q -x 2>&1 | grep -vE 'RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids' | tail -3) > $O/pytest_eval.log 2>&1
tail -2 $O/pytest_eval.log
run() { tag=$1; shift; (env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $EXTRA 2>&1 | tail -1) > $O/bench_$tag.json 2>&1; }
EXTRA="--workload infer --model l" run infer_l_new SY_X=1
EXTRA="--workload infer --model l" run infer_l_prev STREAMYOLO_HIP_LIB=$PREV
EXTRA="--workload infer --model s" run infer_s_new SY_X=1
EXTRA="--workload infer --model s" run infer_s_prev STREAMYOLO_HIP_LIB=$PREV
EXTRA="--workload stream --model l --dtype fp16 --steps 50 --warmup 10 --u8-input 1" run stream_new SY_X=1
EXTRA="--workload stream --model l --dtype fp16 --steps 50 --warmup 10 --u8-input 1" run stream_prev STREAMYOLO_HIP_LIB=$PREV
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],3), d.get('step_ms'))" 2>&1 | cut -c1-300; done
