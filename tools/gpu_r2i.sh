#!/bin/bash
# round-2 stage i: second-generation halo kernel (117/118) and whole-K 1x1 tile kernel (121..123): tests, probes with load ablations, A/B bench
mkdir -p gpurun_out/i
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/i
(timeout 900 python -m pytest tests/test_kernels_conv.py -m gpu -q -x 2>&1 | grep -vE 'RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids' | tail -5) > $O/pytest_conv_kernels.log 2>&1
tail -2 $O/pytest_conv_kernels.log
S3=3,6,10,11,13,14,15
S1=2,4,5,7,9,16,17,18,19,20
(timeout 300 python tools/conv_probe.py --mode stats --shapes $S3 --tiles 115,117,118,113,371,373,627,629,883,885 --reps 7 2>&1 | grep -v amdgpu.ids) > $O/conv_probe_3x3_stats.txt 2>&1
(timeout 300 python tools/conv_probe.py --mode dgrad --shapes $S3 --tiles 115,117,118,113 --reps 7 2>&1 | grep -v amdgpu.ids) > $O/conv_probe_3x3_dgrad.txt 2>&1
(timeout 300 python tools/conv_probe.py --mode stats --shapes $S1 --tiles 86,102,83,120,121,122,123,377,633,889 --reps 7 2>&1 | grep -v amdgpu.ids) > $O/conv_probe_1x1_stats.txt 2>&1
(timeout 300 python tools/conv_probe.py --mode dgrad --shapes $S1 --tiles 86,102,83,120,121,122,123 --reps 7 2>&1 | grep -v amdgpu.ids) > $O/conv_probe_1x1_dgrad.txt 2>&1
(timeout 300 python tools/conv_probe.py --mode fwd --shapes $S1 --tiles 86,102,83,121,122,123 --reps 7 2>&1 | grep -v amdgpu.ids) > $O/conv_probe_1x1_eval.txt 2>&1
cat $O/conv_probe_3x3_stats.txt $O/conv_probe_3x3_dgrad.txt $O/conv_probe_1x1_stats.txt $O/conv_probe_1x1_dgrad.txt $O/conv_probe_1x1_eval.txt
(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_l_new.json 2>&1
(STREAMYOLO_HALO_TILES=112,113,114,115,116 STREAMYOLO_TILE_1X1K= timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_l_old.json 2>&1
(STREAMYOLO_TILE_1X1K= timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_l_halo2_only.json 2>&1
(timeout 300 python bench.py --workload infer --model l --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_infer_l.json 2>&1
(timeout 600 python tools/profile_train.py 2>&1 | grep -v amdgpu.ids) > $O/train_l_layer_profile.txt 2>&1
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],3), d.get('step_ms'), round(d['roofline']['frac'],4), d['roofline'].get('per_kind_ms'))" 2>&1 | cut -c1-600; done
head -75 $O/train_l_layer_profile.txt
