#!/bin/bash
# round-2 final archive: full GPU suite, smoke, bench matrix, rocprof kernel stats (one step), PMC traffic, layer profile
mkdir -p gpurun_out/final
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/final
(timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -vE 'RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids') > $O/pytest_gpu_all.log 2>&1
grep -E "passed|failed" $O/pytest_gpu_all.log | tail -2
(timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2) > $O/smoke.log 2>&1
(timeout 600 python bench.py 2>&1 | tail -1) > $O/bench_train_l_default_run.json 2>&1
(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --path dropin 2>&1 | tail -1) > $O/bench_train_l_dropin.json 2>&1
(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --batch 4 2>&1 | tail -1) > $O/bench_train_l_b4.json 2>&1
(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype fp16 2>&1 | tail -1) > $O/bench_train_l_fp16.json 2>&1
(timeout 300 python bench.py --workload train --model s --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_s.json 2>&1
(timeout 300 python bench.py --workload train --model m --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_m.json 2>&1
(timeout 300 python bench.py --workload infer --model l --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_infer_l.json 2>&1
(timeout 300 python bench.py --workload infer --model s --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_infer_s.json 2>&1
(timeout 300 python bench.py --workload stream --model l --dtype fp16 --steps 50 --warmup 10 --no-cpu-baseline --u8-input 1 2>&1 | tail -1) > $O/bench_stream_l_fp16_u8.json 2>&1
(cd /tmp && rm -rf /tmp/prof_final && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -- python $GRAFT_REPO_ROOT/bench.py --workload train --model l --steps 5 --warmup 3 --no-cpu-baseline 2>&1 | tail -1) > $O/rocprof_bench_line.json 2>&1
cp /tmp/prof_final/*/*kernel_stats.csv $O/train_l_b8_bf16_kernel_stats.csv 2>/dev/null
python tools/trace_analyze.py $(ls /tmp/prof_final/*/*kernel_trace.csv | head -1) > $O/rocprof_last_step.txt 2>&1
(timeout 900 python tools/pmc_traffic.py --out $O/traffic_train_l.json -- --workload train --model l 2>&1 | tail -16) > $O/traffic_train_l.txt 2>&1
(timeout 900 python tools/pmc_traffic.py --out $O/traffic_infer_l.json -- --workload infer --model l 2>&1 | tail -10) > $O/traffic_infer_l.txt 2>&1
(timeout 600 python tools/profile_train.py 2>&1 | grep -v amdgpu.ids) > $O/train_l_layer_profile.txt 2>&1
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],3), d.get('step_ms'), round(d['roofline']['frac'],4), d['roofline'].get('kernel_ms_per_step'), d.get('cpu_baseline'))" 2>&1 | cut -c1-500; done
cat $O/smoke.log; tail -18 $O/traffic_train_l.txt; head -34 $O/rocprof_last_step.txt
