#!/bin/bash
# round-2 stage m: statistics epilogue ablations (no atomics / no cross-lane reduction)
mkdir -p gpurun_out/m
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/m
(timeout 300 python tools/conv_probe.py --mode stats --shapes 9,5,2 --tiles 121,2169,6265,86,2134,6230 --reps 9 --chain 4 2>&1 | grep -v amdgpu.ids) > $O/stats_ablation.txt 2>&1
(timeout 300 python tools/conv_probe.py --mode stats --shapes 10,6,13 --tiles 117,2165,6261,118,2166,6262 --reps 9 --chain 4 2>&1 | grep -v amdgpu.ids) >> $O/stats_ablation.txt 2>&1
(timeout 300 python tools/conv_probe.py --mode dgrad --shapes 9,5,2,10,6,13 --tiles 121,117,118,86 --reps 9 --chain 4 2>&1 | grep -v amdgpu.ids) >> $O/stats_ablation.txt 2>&1
cat $O/stats_ablation.txt
