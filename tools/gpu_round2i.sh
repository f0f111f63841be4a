#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
: > $GRAFT_REPO_ROOT/gpurun_out/pmc_wgrad_summary.txt
PROBE="python $GRAFT_REPO_ROOT/tools/wgrad_probe.py --shapes 10 --reps 2"
(timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_w_a -- $PROBE 2>&1 | tail -1) > /dev/null 2>&1
(timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_w_b -- $PROBE 2>&1 | tail -1) > /dev/null 2>&1
(timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_w_t -- $PROBE 2>&1 | tail -1) > /dev/null 2>&1
for p in a b t; do python $GRAFT_REPO_ROOT/tools/pmc_summary.py $GRAFT_REPO_ROOT/gpurun_out/pmc_w_$p wgrad; done >> $GRAFT_REPO_ROOT/gpurun_out/pmc_wgrad_summary.txt 2>&1
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_w_a $GRAFT_REPO_ROOT/gpurun_out/pmc_w_b $GRAFT_REPO_ROOT/gpurun_out/pmc_w_t
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_wgrad_summary.txt
