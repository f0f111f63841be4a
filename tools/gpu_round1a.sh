#!/bin/bash
# first GPU visit: parity tests, inference bench, rocprof kernel stats, per-layer table
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -60) > gpurun_out/pytest_gpu.log 2>&1
(timeout 300 python bench.py --workload infer --model s --batch 8 --steps 20 --warmup 5 2>&1 | tail -5) > gpurun_out/bench_infer_s.log 2>&1
(timeout 300 python bench.py --workload infer --model l --batch 8 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -5) > gpurun_out/bench_infer_l.log 2>&1
(timeout 300 python tools/profile_layers.py --model l --batch 8 --dtype bf16 2>&1 | tail -200) > gpurun_out/layers_l_bf16.log 2>&1
(timeout 300 python tools/profile_layers.py --model l --batch 8 --dtype fp32 2>&1 | tail -200) > gpurun_out/layers_l_fp32.log 2>&1
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_infer_l -- python $GRAFT_REPO_ROOT/bench.py --workload infer --model l --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --graph 0 2>&1 | tail -5) > gpurun_out/rocprof_infer_l.log 2>&1
find gpurun_out/prof_infer_l -name "*stats*" | head; ls -la gpurun_out
cat gpurun_out/pytest_gpu.log | tail -30
cat gpurun_out/bench_infer_s.log gpurun_out/bench_infer_l.log
