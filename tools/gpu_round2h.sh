#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3) > gpurun_out/pytest_gpu_2h.log 2>&1
cat gpurun_out/pytest_gpu_2h.log
(timeout 600 python tools/profile_train.py 2>&1 | tail -75) > gpurun_out/profile_train_l_2h.log 2>&1
head -50 gpurun_out/profile_train_l_2h.log
