#!/bin/bash
# tools/pmc_kernel.sh STAGE NAME PATTERN -- <python command>
# SQ / LDS counter passes (rocprofv3 --pmc, own runs, kernel-trace only) over one probe command, summarised per kernel
# instance whose name contains PATTERN -> gpurun_out/STAGE/pmc_NAME.txt.  Counters that the box does not list are dropped.
STAGE=$1; NAME=$2; PAT=$3; shift 3; [ "$1" = "--" ] && shift
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/$STAGE
mkdir -p "$O"
ROOT=$PWD
avail=$(cd /tmp && rocprofv3 -L 2>/dev/null | grep -oE "\bSQ_[A-Z0-9_]+|\bGRBM_[A-Z0-9_]+|\bTCP_[A-Z0-9_]+" | sort -u)
pick() { for c in "$@"; do echo "$avail" | grep -qx "$c" && printf "%s " "$c"; done; }
P1=$(pick SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES)
P2=$(pick SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU SQ_INSTS_MFMA)
P3=$(pick SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM)
{
echo "# $*"
i=0
for P in "$P1" "$P2" "$P3"; do
    i=$((i + 1))
    [ -z "$P" ] && continue
    rm -rf /tmp/pmc_${STAGE}_$i
    (cd /tmp && timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmc_${STAGE}_$i -- "$@" > /tmp/pmc_${STAGE}_$i.log 2>&1)
    echo "## pass $i: $P"
    python $ROOT/tools/pmc_summary.py /tmp/pmc_${STAGE}_$i "$PAT"
done
} > $O/pmc_$NAME.txt 2>&1
tail -40 $O/pmc_$NAME.txt
