#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_kernels_conv.py -m gpu -x -q -k every_tile 2>&1 | tail -3) > gpurun_out/pytest_gpu_2f.log 2>&1
cat gpurun_out/pytest_gpu_2f.log
(timeout 600 python tools/conv_probe.py --shapes 3,5,6,7,9,10,11,12,13,14 --tiles 86,102,87,103,83,99 --reps 5 --chain 20 2>&1 | tail -12) > gpurun_out/conv_probe_2f.log 2>&1
cat gpurun_out/conv_probe_2f.log
