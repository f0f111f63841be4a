#!/bin/bash
# round-2 stage r: SQ / LDS counters of the weight-gradient kernels
mkdir -p gpurun_out/r
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r
PA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
PB="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
PD="SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS"
i=0
for P in "$PA" "$PB" "$PD"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_w$i -- python $GRAFT_REPO_ROOT/tools/wgrad_probe.py --shapes 10,9 --variants 65/256,67/256,18/512,17/512 --reps 3 > /dev/null 2>&1)
  echo "# pass $i: $P (wgrad: d4.m.c2 N16 38x60 256->256 3x3 [65, 67], d4.m.c1 1x1 [18, 17])" >> $O/pmc_wgrad.txt
  python tools/pmc_summary.py $O/pmc_w$i wgrad >> $O/pmc_wgrad.txt 2>&1
  rm -rf $O/pmc_w$i
done
cat $O/pmc_wgrad.txt
