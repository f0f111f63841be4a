#!/bin/bash
# round-2 stage u: training tests with 8 statistic replicas (fused finalize), smoke
mkdir -p gpurun_out/u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/u
(STREAMYOLO_STAT_COPIES=8 timeout 1200 python -m pytest tests/test_model_train.py tests/test_optim.py tests/test_amp.py -m gpu -q 2>&1 | grep -vE 'RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids' | tail -12) > $O/pytest_train_copies8.log 2>&1
tail -12 $O/pytest_train_copies8.log
(STREAMYOLO_STAT_COPIES=8 timeout 900 python -m pytest tests/test_model_train.py -m gpu -q -s -k "per_parameter or full_size or m_vs_reference" 2>&1 | grep -iE "median|worst|rel|loss|grad" | head -40) > $O/parity_numbers_copies8.log 2>&1
cat $O/parity_numbers_copies8.log
