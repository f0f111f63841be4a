#!/bin/bash
# round-2 run A: full GPU suite with the new parity prints + bench lines (default, drop-in path, batch 4)
mkdir -p gpurun_out/a
cd $GRAFT_REPO_ROOT
O=gpurun_out/a
(timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -vE 'RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids' | tail -120) > $O/pytest_gpu_all.log 2>&1
(timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 10 2>&1 | tail -1) > $O/bench_train_l.json 2>&1
(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --path dropin 2>&1 | tail -1) > $O/bench_train_l_dropin.json 2>&1
(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --batch 4 2>&1 | tail -1) > $O/bench_train_l_b4.json 2>&1
tail -60 $O/pytest_gpu_all.log
for f in $O/bench_*.json; do echo $f; cut -c1-700 $f; done
