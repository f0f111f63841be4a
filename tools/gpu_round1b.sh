#!/bin/bash
# second GPU visit: training parity on hardware, training bench, rocprof kernel stats (csv)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
nproc; cat /sys/fs/cgroup/cpu.max; python -c "import os; print(len(os.sched_getaffinity(0)))"
(timeout 900 python -m pytest tests/test_model_train.py tests/test_kernels_conv.py -m gpu -x -q -s 2>&1 | tail -40) > gpurun_out/pytest_gpu_train.log 2>&1
(timeout 600 python bench.py --workload train --model l --batch 8 --steps 10 --warmup 3 --cpu-seconds 20 2>&1 | tail -5) > gpurun_out/bench_train_l.log 2>&1
(timeout 300 python bench.py --workload train --model s --batch 8 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -3) > gpurun_out/bench_train_s.log 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_train_l -- python $GRAFT_REPO_ROOT/bench.py --workload train --model l --batch 8 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -3) > gpurun_out/rocprof_train_l.log 2>&1
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_infer_l -- python $GRAFT_REPO_ROOT/bench.py --workload infer --model l --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --graph 0 2>&1 | tail -3) > gpurun_out/rocprof_infer_l.log 2>&1
find gpurun_out -name "*stats*" | head -20
for f in $(find gpurun_out -name "*kernel_stats.csv"); do echo "== $f"; head -12 $f; done
rm -f $(find gpurun_out -name "*kernel_trace.csv") $(find gpurun_out -name "*.db")
tail -25 gpurun_out/pytest_gpu_train.log
cat gpurun_out/bench_train_l.log gpurun_out/bench_train_s.log
