#!/bin/bash
# LDS-DMA conv pipeline: correctness on hardware first (race screen = repeated runs), then speed
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/pytest_gpu_d.log 2>&1
(timeout 600 python -m pytest tests/test_kernels_conv.py tests/test_model_eval.py -m gpu -x -q --count 1 2>&1 | tail -3) >> gpurun_out/pytest_gpu_d.log 2>&1
(timeout 300 python tools/profile_layers.py --model l --batch 8 --dtype bf16 2>&1 | tail -200) > gpurun_out/layers_l_bf16_d.log 2>&1
(timeout 300 python bench.py --workload infer --model l --batch 8 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_infer_l_d.log 2>&1
(timeout 300 python bench.py --workload infer --model s --batch 8 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_infer_s_d.log 2>&1
(timeout 600 python bench.py --workload train --model l --batch 8 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_train_l_d.log 2>&1
tail -6 gpurun_out/pytest_gpu_d.log
tail -3 gpurun_out/layers_l_bf16_d.log
cat gpurun_out/bench_infer_l_d.log gpurun_out/bench_train_l_d.log | cut -c1-1500
