#!/bin/bash
# round-2 run B: GPU suite (full log) + bench lines after the BatchNorm fusion step 1 + one-step rocprof summary
mkdir -p gpurun_out/b
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/b
(timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -vE 'RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids') > $O/pytest_gpu_all.log 2>&1
(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_l.json 2>&1
(STREAMYOLO_FUSE_REDUCE=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_l_nofuse.json 2>&1
(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --batch 4 2>&1 | tail -1) > $O/bench_train_l_b4.json 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --workload train --model l --steps 5 --warmup 3 --no-cpu-baseline 2>&1 | tail -1) > $O/rocprof_bench_line.json 2>&1
cp $O/prof/*/*kernel_stats.csv $O/train_l_b8_bf16_kernel_stats.csv 2>/dev/null
rm -rf $O/prof
grep -E "passed|failed" $O/pytest_gpu_all.log | tail -3
grep -E "rel err|agreement|per-parameter|anchors the oracle" $O/pytest_gpu_all.log | grep -v print | head -40
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],2), d['step_ms'], d['config']['host_launch_ms_per_step'], d['roofline']['per_kind_ms'], round(d['roofline']['frac'],4))"; done
head -25 $O/train_l_b8_bf16_kernel_stats.csv | cut -c1-150
