#!/bin/bash
# round-2 stage k: what do the BatchNorm-statistics atomics of the conv epilogue cost? (replica count sweep)
mkdir -p gpurun_out/k
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/k
for c in 1 4 16 32 64 256; do
  echo "== copies $c" >> $O/stats_copies.txt
  (timeout 300 python tools/conv_probe.py --mode stats --copies $c --shapes 9,5,10,13,2 --tiles 121,117,86,115 --reps 9 --chain 4 2>&1 | grep -v amdgpu.ids) >> $O/stats_copies.txt 2>&1
done
echo "== dgrad (no statistics)" >> $O/stats_copies.txt
(timeout 300 python tools/conv_probe.py --mode dgrad --shapes 9,5,10,13,2 --tiles 121,117,86,115 --reps 9 --chain 4 2>&1 | grep -v amdgpu.ids) >> $O/stats_copies.txt 2>&1
cat $O/stats_copies.txt
