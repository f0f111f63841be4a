#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > gpurun_out/pytest_gpu_i.log 2>&1
(timeout 600 python bench.py --workload train --model l --steps 10 --warmup 3 --cpu-seconds 15 2>&1 | tail -1) > gpurun_out/bench_train_l_i.log 2>&1
(timeout 300 python bench.py --workload infer --model l --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_infer_l_i.log 2>&1
(timeout 300 python bench.py --workload infer --model s --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_infer_s_i.log 2>&1
(timeout 300 python bench.py --workload stream --model l --dtype fp16 --steps 50 --warmup 10 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_stream_l_i.log 2>&1
(timeout 300 python bench.py --workload train --model s --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_train_s_i.log 2>&1
tail -3 gpurun_out/pytest_gpu_i.log
for f in gpurun_out/bench_*_i.log; do echo $f; cut -c1-400 $f; done
