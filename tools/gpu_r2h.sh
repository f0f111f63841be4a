#!/bin/bash
mkdir -p gpurun_out/h
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/h
(timeout 900 python -m pytest tests/test_kernels_conv.py tests/test_data.py -m gpu -q -x 2>&1 | tail -3) > $O/pytest_some.log 2>&1
tail -2 $O/pytest_some.log
(timeout 600 python tools/conv_probe.py --mode dgrad --shapes 2,4,5,7,9 --tiles 86,102,83,22,120 --reps 7 --chain 10 2>&1 | tail -7) > $O/conv_probe_1x1_dgrad.txt 2>&1
cat $O/conv_probe_1x1_dgrad.txt
run() { tag=$1; shift; (env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_$tag.json 2>&1
  echo "== $tag $@"; python -c "
import json,sys
d=json.load(open('$O/bench_$tag.json')); print(round(d['value'],1), round(d['ms_per_step'],2), d['config']['host_launch_ms_per_step'], d['roofline']['per_kind_ms'], round(d['roofline']['frac'],4))"; }
run stream X=1
run nostream STREAMYOLO_STREAM_1X1=0
(timeout 600 python tools/profile_train.py 2>&1 | grep -v amdgpu.ids) > $O/train_l_layer_profile.txt 2>&1
grep -E "k1 s1" $O/train_l_layer_profile.txt | head -40
