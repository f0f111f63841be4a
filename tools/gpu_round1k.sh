#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > gpurun_out/pytest_gpu_k.log 2>&1
(timeout 600 python bench.py --workload train --model l --steps 10 --warmup 3 --cpu-seconds 15 2>&1 | tail -1) > gpurun_out/bench_train_l_k.log 2>&1
(timeout 300 python bench.py --workload infer --model l --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_infer_l_k.log 2>&1
(timeout 300 python bench.py --workload train --model s --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_train_s_k.log 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_train_l_k -- python $GRAFT_REPO_ROOT/bench.py --workload train --model l --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/rocprof_train_l_k.log 2>&1
(timeout 500 python tools/conv_probe.py --tiles 86,88,89,24,83 2>&1 | tail -40) > gpurun_out/conv_probe_k.log 2>&1
rm -f $(find gpurun_out -name "*kernel_trace.csv") $(find gpurun_out -name "*.db")
tail -3 gpurun_out/pytest_gpu_k.log
for f in gpurun_out/bench_*_k.log; do echo $f; cut -c1-300 $f; done
head -14 gpurun_out/prof_train_l_k/*/*kernel_stats.csv | cut -c1-160
cat gpurun_out/conv_probe_k.log
