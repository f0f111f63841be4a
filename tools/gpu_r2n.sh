#!/bin/bash
# round-2 stage n: DPP statistics reduction in the conv epilogue: kernel tests, probes, A/B bench is vs stage l (same tree otherwise)
mkdir -p gpurun_out/n
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/n
(timeout 900 python -m pytest tests/test_kernels_conv.py tests/test_kernels_misc.py -m gpu -q -x 2>&1 | grep -vE 'RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids' | tail -4) > $O/pytest_kernels.log 2>&1
tail -2 $O/pytest_kernels.log
(timeout 300 python tools/conv_probe.py --mode stats --shapes 9,5,2 --tiles 121,2169,6265,86,120 --reps 9 --chain 4 2>&1 | grep -v amdgpu.ids) > $O/stats_dpp.txt 2>&1
(timeout 300 python tools/conv_probe.py --mode stats --shapes 10,6,13 --tiles 117,2165,6261,118,115 --reps 9 --chain 4 2>&1 | grep -v amdgpu.ids) >> $O/stats_dpp.txt 2>&1
cat $O/stats_dpp.txt
(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_l.json 2>&1
(timeout 300 python bench.py --workload train --model s --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_s.json 2>&1
(timeout 600 python tools/profile_train.py 2>&1 | grep -v amdgpu.ids) > $O/train_l_layer_profile.txt 2>&1
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],3), d.get('step_ms'), round(d['roofline']['frac'],4), d['roofline'].get('per_kind_ms'))" 2>&1 | cut -c1-600; done
head -60 $O/train_l_layer_profile.txt
