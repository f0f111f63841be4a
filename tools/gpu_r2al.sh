#!/bin/bash
# round-2 stage al: rows in flight per thread in the BatchNorm row kernels (2 / 3 / 4)
mkdir -p gpurun_out/al
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/al
L=$GRAFT_REPO_ROOT/streamyolo_amd/lib
(STREAMYOLO_HIP_LIB=$L/ab_d4.so timeout 600 python -m pytest tests/test_kernels_misc.py -m gpu -q -x -k "bn or segmented" 2>&1 | tail -1) > $O/pytest_d4.log 2>&1
cat $O/pytest_d4.log
run() { tag=$1; shift; (env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_l_$tag.json 2>&1; }
run d2 STREAMYOLO_HIP_LIB=$L/ab_d2.so
run d3 STREAMYOLO_HIP_LIB=$L/ab_d3.so
run d4 STREAMYOLO_HIP_LIB=$L/ab_d4.so
run d2b STREAMYOLO_HIP_LIB=$L/ab_d2.so
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],3), {k: v for k, v in d['roofline'].get('per_kind_ms').items() if k.startswith('bn_')})" 2>&1 | cut -c1-300; done
