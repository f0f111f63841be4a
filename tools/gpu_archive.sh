#!/bin/bash
# stage archive (copy the outputs to profiles/rNN/<stage>_*): full GPU suite, bench matrix, kernel stats, PMC traffic, probes
mkdir -p gpurun_out/f
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/f
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -vE 'RCCL|HIP version|ROCm version|Hostname|Librccl' | tail -8) > $O/pytest_gpu_all.log 2>&1
(timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2) > $O/smoke.log 2>&1
(timeout 600 python bench.py --workload train --model l --steps 20 --warmup 5 --cpu-seconds 15 2>&1 | tail -1) > $O/bench_train_l.json 2>&1
(timeout 300 python bench.py --workload train --model s --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_s.json 2>&1
(timeout 300 python bench.py --workload infer --model l --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_infer_l.json 2>&1
(timeout 300 python bench.py --workload infer --model s --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_infer_s.json 2>&1
(timeout 300 python bench.py --workload stream --model l --dtype fp16 --steps 50 --warmup 10 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_stream_l_fp16.json 2>&1
(timeout 300 python bench.py --workload stream --model l --dtype fp16 --steps 50 --warmup 10 --no-cpu-baseline --u8-input 1 2>&1 | tail -1) > $O/bench_stream_l_fp16_u8.json 2>&1
(timeout 300 python bench.py --workload stream --model l --dtype fp16 --steps 50 --warmup 10 --no-cpu-baseline --u8-input 1 --h2d 1 2>&1 | tail -1) > $O/bench_stream_l_fp16_u8_h2d.json 2>&1
(timeout 300 python bench.py --workload stream --model l --dtype fp16 --steps 50 --warmup 10 --no-cpu-baseline --h2d 1 2>&1 | tail -1) > $O/bench_stream_l_fp16_h2d.json 2>&1
(timeout 300 python bench.py --workload train --model m --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_m.json 2>&1
(timeout 300 python bench.py --workload train --model l --steps 20 --warmup 5 --no-cpu-baseline --with-optimizer 1 2>&1 | tail -1) > $O/bench_train_l_with_optimizer.json 2>&1
(timeout 300 python tools/host_profile.py l 2>&1 | grep -v "^$" | tail -32) > $O/host_profile_train_l.txt 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --workload train --model l --steps 5 --warmup 3 --no-cpu-baseline 2>&1 | tail -1) > $O/rocprof_bench_line.json 2>&1
cp $O/prof/*/*kernel_stats.csv $O/train_l_b8_bf16_kernel_stats.csv
rm -rf $O/prof
(timeout 900 python tools/pmc_traffic.py --out $O/traffic_train_l.json -- --workload train --model l 2>&1 | tail -14) > $O/traffic_train_l.txt 2>&1
(timeout 900 python tools/pmc_traffic.py --out $O/traffic_infer_l.json -- --workload infer --model l 2>&1 | tail -8) > $O/traffic_infer_l.txt 2>&1
(timeout 600 python tools/profile_train.py 2>&1 | tail -75) > $O/train_l_layer_profile.txt 2>&1
(timeout 600 python tools/conv_probe.py --tiles 86,102,87,103,83,22,19,38 --reps 5 --chain 10 2>&1 | tail -18) > $O/conv_variants_probe.txt 2>&1
(timeout 600 python tools/wgrad_probe.py 2>&1 | tail -18) > $O/wgrad_variants_probe.txt 2>&1
cat $O/pytest_gpu_all.log
for f in $O/bench_*.json; do echo $f; cut -c1-260 $f; done
cat $O/smoke.log $O/traffic_train_l.txt
