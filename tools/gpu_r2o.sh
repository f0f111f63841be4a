#!/bin/bash
# round-2 stage o: hardware bf16 conversion + v_rcp sigmoid + two rows in flight in the BatchNorm row kernels: GPU tests (all), bench
mkdir -p gpurun_out/o
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/o
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -vE 'RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids' | tail -8) > $O/pytest_gpu_all.log 2>&1
tail -4 $O/pytest_gpu_all.log
(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_l.json 2>&1
(timeout 300 python bench.py --workload train --model s --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_s.json 2>&1
(timeout 300 python bench.py --workload infer --model l --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_infer_l.json 2>&1
(timeout 300 python bench.py --workload stream --model l --dtype fp16 --steps 50 --warmup 10 --no-cpu-baseline --u8-input 1 2>&1 | tail -1) > $O/bench_stream_l_fp16_u8.json 2>&1
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],3), d.get('step_ms'), round(d['roofline']['frac'],4), d['roofline'].get('per_kind_ms'))" 2>&1 | cut -c1-600; done
