#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 600 python tools/conv_probe.py --shapes 3,6,10,13,15 --tiles 19,275,531,787,23,279,535,791 2>&1 | tail -12) > gpurun_out/conv_ablate.log 2>&1
(timeout 600 python -m pytest tests/test_model_train.py tests/test_kernels_misc.py -m gpu -x -q 2>&1 | tail -3) > gpurun_out/pytest_gpu_g.log 2>&1
(timeout 600 python bench.py --workload train --model l --batch 8 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_train_l_g.log 2>&1
cat gpurun_out/conv_ablate.log gpurun_out/pytest_gpu_g.log; cut -c1-300 gpurun_out/bench_train_l_g.log
