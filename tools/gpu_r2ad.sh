#!/bin/bash
# round-2 stage ad: kernel summary of the inference step (l, 8 pairs) and of the streaming frame
mkdir -p gpurun_out/ad
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/ad
(cd /tmp && rm -rf /tmp/prof_i && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_i -- python $GRAFT_REPO_ROOT/bench.py --workload infer --model l --steps 10 --warmup 3 --no-cpu-baseline --graph 0 > /dev/null 2>&1)
cp /tmp/prof_i/*/*kernel_stats.csv $O/infer_l_kernel_stats.csv
head -25 $O/infer_l_kernel_stats.csv | cut -c1-200
(cd /tmp && rm -rf /tmp/prof_s && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -- python $GRAFT_REPO_ROOT/bench.py --workload stream --model l --dtype fp16 --steps 30 --warmup 5 --no-cpu-baseline --u8-input 1 --graph 0 > /dev/null 2>&1)
cp /tmp/prof_s/*/*kernel_stats.csv $O/stream_l_kernel_stats.csv
head -25 $O/stream_l_kernel_stats.csv | cut -c1-200
