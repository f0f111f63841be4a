#!/bin/bash
# one-off diagnosis: the single-rank RCCL bucket test under the stream-split switches
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/i
for v in "STREAMYOLO_BWD_SPLIT_FRAMES=0 STREAMYOLO_FWD_SPLIT_FRAMES=1" "STREAMYOLO_BWD_SPLIT_FRAMES=1 STREAMYOLO_FWD_SPLIT_FRAMES=0" "STREAMYOLO_BWD_SPLIT_FRAMES=0 STREAMYOLO_FWD_SPLIT_FRAMES=0" "STREAMYOLO_BWD_SPLIT_FRAMES=1 STREAMYOLO_FWD_SPLIT_FRAMES=1"; do
  echo "== $v"
  env $v timeout 300 python -m pytest tests/test_distributed_gloo.py -q -m gpu -k single_rank 2>&1 | grep -E "passed|failed|AssertionError: \(" | tail -3
done
