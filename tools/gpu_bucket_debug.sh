#!/bin/bash
# one-off diagnosis: the single-rank RCCL bucket test in suite order, repeated
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/i
for r in 1 2 3; do
  timeout 300 python -m pytest tests/test_abi.py tests/test_amp.py tests/test_data.py tests/test_distributed_gloo.py -q -m gpu 2>&1 | grep -E "passed|failed|AssertionError: \(" | tail -3
done
for r in 1 2 3 4 5 6; do
  timeout 300 python -m pytest tests/test_distributed_gloo.py -q -m gpu -k single_rank 2>&1 | grep -E "passed|failed|AssertionError: \(" | tail -2
done
