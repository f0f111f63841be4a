#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
: > $GRAFT_REPO_ROOT/gpurun_out/pmc_1x1_summary.txt
for T in 86 854 83; do
PROBE="python $GRAFT_REPO_ROOT/tools/conv_probe.py --shapes 10 --tiles $T --reps 3"
(timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_1x1_a -- $PROBE 2>&1 | tail -1) > /dev/null 2>&1
(timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_1x1_b -- $PROBE 2>&1 | tail -1) > /dev/null 2>&1
(timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_1x1_t -- $PROBE 2>&1 | tail -1) > /dev/null 2>&1
echo "== tile $T" >> $GRAFT_REPO_ROOT/gpurun_out/pmc_1x1_summary.txt
for p in a b t; do python $GRAFT_REPO_ROOT/tools/pmc_summary.py $GRAFT_REPO_ROOT/gpurun_out/pmc_1x1_$p; done >> $GRAFT_REPO_ROOT/gpurun_out/pmc_1x1_summary.txt 2>&1
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_1x1_a $GRAFT_REPO_ROOT/gpurun_out/pmc_1x1_b $GRAFT_REPO_ROOT/gpurun_out/pmc_1x1_t
done
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_1x1_summary.txt
