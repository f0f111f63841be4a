#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_kernels_conv.py tests/test_model_eval.py tests/test_model_train.py -m gpu -x -q 2>&1 | tail -4) > gpurun_out/pytest_gpu_f.log 2>&1
(timeout 600 python tools/conv_probe.py 2>&1 | tail -30) > gpurun_out/conv_probe_f.log 2>&1
(timeout 300 python bench.py --workload infer --model l --batch 8 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_infer_l_f.log 2>&1
(timeout 300 python bench.py --workload infer --model s --batch 8 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_infer_s_f.log 2>&1
(timeout 600 python bench.py --workload train --model l --batch 8 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_train_l_f.log 2>&1
cd /tmp
PROBE="python $GRAFT_REPO_ROOT/tools/conv_probe.py --shapes 13 --tiles 19 --reps 3"
(timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_f_b -- $PROBE 2>&1 | tail -1) > /dev/null 2>&1
(timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_f_c -- $PROBE 2>&1 | tail -1) > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
rm -f $(find gpurun_out -name "*.db")
cat gpurun_out/pytest_gpu_f.log; cat gpurun_out/conv_probe_f.log
cat gpurun_out/bench_infer_l_f.log gpurun_out/bench_train_l_f.log | cut -c1-300
