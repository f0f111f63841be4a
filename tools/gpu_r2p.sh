#!/bin/bash
# round-2 stage p: one-step kernel summary (rocprofv3 kernel trace) at batch 8 and batch 4
mkdir -p gpurun_out/p
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/p
for B in 8 4; do
  (cd /tmp && rm -rf /tmp/trace_out && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_out -- python $GRAFT_REPO_ROOT/bench.py --workload train --model l --batch $B --steps 5 --warmup 3 --no-cpu-baseline 2>&1 | tail -1) > $O/rocprof_bench_line_b$B.json 2>&1
  python tools/trace_analyze.py $(ls /tmp/trace_out/*/*kernel_trace.csv | head -1) > $O/rocprof_last_step_b$B.txt 2>&1
  cp /tmp/trace_out/*/*kernel_stats.csv $O/train_l_b${B}_kernel_stats.csv 2>/dev/null
done
head -45 $O/rocprof_last_step_b8.txt; head -45 $O/rocprof_last_step_b4.txt
