#!/bin/bash
# round-2 stage aj: BatchNorm grid caps, second sweep (fewer, fatter workgroups)
mkdir -p gpurun_out/aj
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/aj
run() { tag=$1; shift; (env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_l_$tag.json 2>&1; }
run base SY_X=1
run bapply512 SY_BN_BAPPLY_BLOCKS=512
run bapply768 SY_BN_BAPPLY_BLOCKS=768
run apply1024 SY_BN_APPLY_BLOCKS=1024
run apply1536 SY_BN_APPLY_BLOCKS=1536
run reduce512 SY_BN_REDUCE_BLOCKS=512
run combo_a SY_BN_BAPPLY_BLOCKS=1024 SY_BN_APPLY_BLOCKS=2048 SY_BN_REDUCE_BLOCKS=768
run combo_b SY_BN_BAPPLY_BLOCKS=768 SY_BN_APPLY_BLOCKS=1536 SY_BN_REDUCE_BLOCKS=768
run base2 SY_X=1
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],3), {k: v for k, v in d['roofline'].get('per_kind_ms').items() if k.startswith('bn_')})" 2>&1 | cut -c1-300; done
