#!/bin/bash
# round-2 run C: halo kernel on hardware (tests + per-layer probe fwd / dgrad), g-space v2, bench lines
mkdir -p gpurun_out/c
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/c
(timeout 900 python -m pytest tests/test_kernels_conv.py tests/test_kernels_misc.py -m gpu -q -x 2>&1 | grep -vE 'RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids' | tail -15) > $O/pytest_kernels.log 2>&1
cat $O/pytest_kernels.log | tail -5
(timeout 600 python tools/conv_probe.py --shapes 3,6,10,11,13,14,15 --tiles 86,102,83,99,112,113,114,115,116 --reps 7 --chain 10 2>&1 | tail -9) > $O/conv_probe_fwd.txt 2>&1
cat $O/conv_probe_fwd.txt
(timeout 600 python tools/conv_probe.py --mode dgrad --shapes 3,6,10,11,13,14,15 --tiles 86,102,83,99,112,113,114,115,116 --reps 7 --chain 10 2>&1 | tail -9) > $O/conv_probe_dgrad.txt 2>&1
cat $O/conv_probe_dgrad.txt
(timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -vE 'RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids') > $O/pytest_gpu_all.log 2>&1
grep -E "passed|failed" $O/pytest_gpu_all.log | tail -3
grep -E "rel err|agreement|per-parameter|anchors the oracle" $O/pytest_gpu_all.log | grep -v print | head -40
for v in "" "STREAMYOLO_FUSE_REDUCE=0" "STREAMYOLO_HALO_TILES=" "STREAMYOLO_HALO_TILES= STREAMYOLO_FUSE_REDUCE=0"; do
  tag=$(echo "$v" | tr -c 'A-Za-z0-9' '_')
  (env $v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_l_$tag.json 2>&1
  echo "== $v"; python -c "
import json,sys
d=json.load(open('$O/bench_train_l_$tag.json')); print(round(d['value'],1), round(d['ms_per_step'],2), d['step_ms'], d['config']['host_launch_ms_per_step'], d['roofline']['per_kind_ms'], round(d['roofline']['frac'],4))"
done
