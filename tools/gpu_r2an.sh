#!/bin/bash
# round-2 stage an: HIP runtime knobs for kernel start latency (kernel arguments in device memory)
mkdir -p gpurun_out/an
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/an
run() { tag=$1; shift; (env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $EXTRA 2>&1 | tail -1) > $O/bench_$tag.json 2>&1; }
run train_l_base SY_X=1
run train_l_devkernarg HIP_FORCE_DEV_KERNARG=1
run train_l_base2 SY_X=1
run train_l_devkernarg2 HIP_FORCE_DEV_KERNARG=1
EXTRA="--batch 4" run train_l_b4_base SY_X=1
EXTRA="--batch 4" run train_l_b4_devkernarg HIP_FORCE_DEV_KERNARG=1
EXTRA="--workload stream --model l --dtype fp16 --steps 50 --warmup 10 --u8-input 1" run stream_base SY_X=1
EXTRA="--workload stream --model l --dtype fp16 --steps 50 --warmup 10 --u8-input 1" run stream_devkernarg HIP_FORCE_DEV_KERNARG=1
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],3), d['config'].get('host_launch_ms_per_step'))" 2>&1 | cut -c1-300; done
