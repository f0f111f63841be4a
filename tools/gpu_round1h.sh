#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 600 python tools/conv_probe.py --shapes 6,13 --tiles 22,278,534,790,19,787 2>&1 | tail -5) > gpurun_out/conv_ablate2.log 2>&1
cd /tmp
PROBE="python $GRAFT_REPO_ROOT/tools/conv_probe.py --shapes 13 --tiles 22,19,790 --reps 3"
(timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_h_a -- $PROBE 2>&1 | tail -1) > /dev/null 2>&1
(timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_h_b -- $PROBE 2>&1 | tail -1) > /dev/null 2>&1
(timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_h_c -- $PROBE 2>&1 | tail -1) > /dev/null 2>&1
(timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_h_d -- $PROBE 2>&1 | tail -1) > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
rm -f $(find gpurun_out -name "*.db")
cat gpurun_out/conv_ablate2.log
for d in pmc_h_a pmc_h_b pmc_h_c pmc_h_d; do ls gpurun_out/$d/*/ | head -3; done
