#!/bin/bash
mkdir -p gpurun_out/e
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/e
(timeout 900 python -m pytest tests/test_kernels_conv.py tests/test_kernels_misc.py tests/test_model_train.py -m gpu -q -x 2>&1 | tail -3) > $O/pytest_some.log 2>&1
tail -2 $O/pytest_some.log
(timeout 600 python tools/conv_probe.py --shapes 3,6,10,11,13,14,15 --tiles 102,83,112,113,114,115,116 --reps 7 --chain 10 2>&1 | tail -8) > $O/conv_probe_fwd.txt 2>&1
cat $O/conv_probe_fwd.txt
run() { tag=$1; shift; (env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_$tag.json 2>&1
  echo "== $tag $@"; python -c "
import json,sys
d=json.load(open('$O/bench_$tag.json')); print(round(d['value'],1), round(d['ms_per_step'],2), d['config']['host_launch_ms_per_step'], d['roofline']['per_kind_ms'], round(d['roofline']['frac'],4))"; }
run base X=1
run atom64k SY_BN_REDUCE_ATOMICS=65536
run atom128k SY_BN_REDUCE_ATOMICS=131072
run rows2 SY_BN_MIN_ROWS=2
run rows4 SY_BN_MIN_ROWS=4
run rows4atom SY_BN_MIN_ROWS=4 SY_BN_REDUCE_ATOMICS=98304
