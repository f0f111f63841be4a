#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_kernels_conv.py tests/test_kernels_misc.py -m gpu -x -q 2>&1 | tail -3) > gpurun_out/pytest_gpu_j.log 2>&1
(timeout 600 python tools/conv_probe.py 2>&1 | tail -18) > gpurun_out/conv_probe_j.log 2>&1
(timeout 600 python tools/wgrad_probe.py 2>&1 | tail -18) > gpurun_out/wgrad_probe_j.log 2>&1
cat gpurun_out/pytest_gpu_j.log gpurun_out/conv_probe_j.log gpurun_out/wgrad_probe_j.log
