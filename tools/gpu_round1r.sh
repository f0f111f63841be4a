#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_train_l -- python $GRAFT_REPO_ROOT/bench.py --workload train --model l --steps 3 --warmup 4 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/trace_train_l.log 2>&1
ls -la gpurun_out/trace_train_l/*/
python tools/trace_analyze.py gpurun_out/trace_train_l/*/*kernel_trace.csv > gpurun_out/trace_analysis.txt 2>&1
cat gpurun_out/trace_analysis.txt
rm -f $(find gpurun_out -name "*kernel_trace.csv") $(find gpurun_out -name "*.db")
