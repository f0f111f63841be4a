#!/bin/bash
# conv kernel variants: correctness of every tile on hardware, timing table, PMC counters on one layer
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_kernels_conv.py tests/test_model_eval.py -m gpu -x -q 2>&1 | tail -4) > gpurun_out/pytest_gpu_e.log 2>&1
(timeout 600 python tools/conv_probe.py 2>&1 | tail -30) > gpurun_out/conv_probe.log 2>&1
cd /tmp
PROBE="python $GRAFT_REPO_ROOT/tools/conv_probe.py --shapes 13 --tiles 1,3,19 --reps 3"
(timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_a -- $PROBE 2>&1 | tail -3) > $GRAFT_REPO_ROOT/gpurun_out/pmc_a.log 2>&1
(timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_b -- $PROBE 2>&1 | tail -3) > $GRAFT_REPO_ROOT/gpurun_out/pmc_b.log 2>&1
(timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_c -- $PROBE 2>&1 | tail -3) > $GRAFT_REPO_ROOT/gpurun_out/pmc_c.log 2>&1
(timeout 300 rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_d -- $PROBE 2>&1 | tail -3) > $GRAFT_REPO_ROOT/gpurun_out/pmc_d.log 2>&1
cd $GRAFT_REPO_ROOT
rm -f $(find gpurun_out -name "*.db")
find gpurun_out/pmc_a gpurun_out/pmc_b gpurun_out/pmc_c gpurun_out/pmc_d -type f | head -20
cat gpurun_out/pytest_gpu_e.log; cat gpurun_out/conv_probe.log
