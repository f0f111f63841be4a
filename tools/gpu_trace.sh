#!/bin/bash
# kernel timeline of the l training step: union-busy vs idle (is the GPU waiting for the host?)
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/trace_out
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_out -- python $GRAFT_REPO_ROOT/bench.py --workload train --model l --steps 4 --warmup 4 --no-cpu-baseline $EXTRA 2>&1 | grep '^{"metric' | tail -1 > $GRAFT_REPO_ROOT/gpurun_out/trace_bench_line.json
f=$(ls /tmp/trace_out/*/*kernel_trace.csv | head -1)
python $GRAFT_REPO_ROOT/tools/trace_analyze.py $f 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/trace_summary.txt
