#!/bin/bash
# third GPU visit: full gpu suite after the training-kernel rewrite, training bench + rocprof
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/pytest_gpu_c.log 2>&1
(timeout 600 python bench.py --workload train --model l --batch 8 --steps 10 --warmup 3 --cpu-seconds 15 2>&1 | tail -2) > gpurun_out/bench_train_l_c.log 2>&1
(timeout 300 python bench.py --workload train --model s --batch 8 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -2) > gpurun_out/bench_train_s_c.log 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_train_l_c -- python $GRAFT_REPO_ROOT/bench.py --workload train --model l --batch 8 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -2) > gpurun_out/rocprof_train_l_c.log 2>&1
rm -f $(find gpurun_out -name "*kernel_trace.csv") $(find gpurun_out -name "*.db")
for f in $(find gpurun_out/prof_train_l_c -name "*kernel_stats.csv"); do echo "== $f"; head -14 $f | cut -c1-220; done
tail -8 gpurun_out/pytest_gpu_c.log
cat gpurun_out/bench_train_l_c.log gpurun_out/bench_train_s_c.log
