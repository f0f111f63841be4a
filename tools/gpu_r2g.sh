#!/bin/bash
# round-2 stage g archive: full GPU suite, smoke, bench matrix, rocprof kernel stats, PMC traffic, layer profile
mkdir -p gpurun_out/g
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/g
(timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -vE 'RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids') > $O/pytest_gpu_all.log 2>&1
grep -E "passed|failed" $O/pytest_gpu_all.log | tail -2
(timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2) > $O/smoke.log 2>&1
(timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 12 2>&1 | tail -1) > $O/bench_train_l.json 2>&1
(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --path dropin 2>&1 | tail -1) > $O/bench_train_l_dropin.json 2>&1
(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --batch 4 2>&1 | tail -1) > $O/bench_train_l_b4.json 2>&1
(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype fp16 2>&1 | tail -1) > $O/bench_train_l_fp16.json 2>&1
(timeout 300 python bench.py --workload train --model s --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_s.json 2>&1
(timeout 300 python bench.py --workload train --model m --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_m.json 2>&1
(timeout 300 python bench.py --workload infer --model l --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_infer_l.json 2>&1
(timeout 300 python bench.py --workload stream --model l --dtype fp16 --steps 50 --warmup 10 --no-cpu-baseline --u8-input 1 2>&1 | tail -1) > $O/bench_stream_l_fp16_u8.json 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --workload train --model l --steps 5 --warmup 3 --no-cpu-baseline 2>&1 | tail -1) > $O/rocprof_bench_line.json 2>&1
cp $O/prof/*/*kernel_stats.csv $O/train_l_b8_bf16_kernel_stats.csv 2>/dev/null
python tools/trace_analyze.py $(ls $O/prof/*/*kernel_trace.csv | head -1) > $O/rocprof_last_step.txt 2>&1
rm -rf $O/prof
(timeout 900 python tools/pmc_traffic.py --out $O/traffic_train_l.json -- --workload train --model l 2>&1 | tail -14) > $O/traffic_train_l.txt 2>&1
(timeout 600 python tools/profile_train.py 2>&1 | grep -v amdgpu.ids) > $O/train_l_layer_profile.txt 2>&1
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],3), d.get('step_ms'), d['config']['host_launch_ms_per_step'], round(d['roofline']['frac'],4), d['roofline']['per_kind_ms'])" 2>&1 | cut -c1-600; done
cat $O/smoke.log; tail -16 $O/traffic_train_l.txt; head -30 $O/rocprof_last_step.txt
