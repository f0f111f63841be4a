#!/bin/bash
# round-2 stage ae: MFMA-busy counters per kernel family over the training step
mkdir -p gpurun_out/ae
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/ae
(cd /tmp && rm -rf /tmp/pmc_m && timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_m -- python $GRAFT_REPO_ROOT/bench.py --workload train --model l --steps 2 --warmup 3 --no-cpu-baseline > /dev/null 2>&1)
python tools/pmc_mfma_util.py /tmp/pmc_m > $O/mfma_util_train_l.txt 2>&1
cat $O/mfma_util_train_l.txt
