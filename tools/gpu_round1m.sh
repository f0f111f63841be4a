#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_model_train.py tests/test_kernels_misc.py -m gpu -x -q 2>&1 | tail -8) > gpurun_out/pytest_gpu_m.log 2>&1
(timeout 600 python bench.py --workload train --model l --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_train_l_m.log 2>&1
(STREAMYOLO_STREAMS=1 timeout 600 python bench.py --workload train --model l --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_train_l_m1.log 2>&1
(timeout 300 python bench.py --workload train --model s --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_train_s_m.log 2>&1
cat gpurun_out/pytest_gpu_m.log
for f in gpurun_out/bench_*_m*.log; do echo $f; cut -c1-240 $f; done
