#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 900 python tools/pmc_traffic.py --out gpurun_out/traffic_train_l.json -- --workload train --model l 2>&1 | tail -20) > gpurun_out/traffic_train_l.log 2>&1
cat gpurun_out/traffic_train_l.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_train_l_x -- python $GRAFT_REPO_ROOT/bench.py --workload train --model l --steps 5 --warmup 3 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/rocprof_train_l_x.log 2>&1
rm -f $(find gpurun_out -name "*kernel_trace.csv") $(find gpurun_out -name "*.db")
head -30 gpurun_out/prof_train_l_x/*/*kernel_stats.csv | cut -c1-200
