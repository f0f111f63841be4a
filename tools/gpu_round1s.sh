#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_kernels_conv.py -m gpu -x -q -k wgrad 2>&1 | tail -5) > gpurun_out/pytest_gpu_s.log 2>&1
(timeout 900 python tools/wgrad_probe.py 2>&1 | tail -20) > gpurun_out/wgrad_probe_s.log 2>&1
cat gpurun_out/pytest_gpu_s.log gpurun_out/wgrad_probe_s.log
