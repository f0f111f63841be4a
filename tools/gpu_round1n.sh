#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_model_train.py -m gpu -x -q -k overlapped 2>&1 | tail -60) > gpurun_out/pytest_gpu_n.log 2>&1
cat gpurun_out/pytest_gpu_n.log
