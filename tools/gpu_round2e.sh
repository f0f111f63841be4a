#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_distributed_gloo.py tests/test_model_train.py -m gpu -x -q 2>&1 | tail -25) > gpurun_out/pytest_gpu_2e.log 2>&1
cat gpurun_out/pytest_gpu_2e.log
(timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 4 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-300) > gpurun_out/bench_torchrun1.log 2>&1
cat gpurun_out/bench_torchrun1.log
