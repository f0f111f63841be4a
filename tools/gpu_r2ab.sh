#!/bin/bash
# round-2 stage ab: does the isolated-kernel tuner pick what is best IN SITU?  (force single variants per family)
mkdir -p gpurun_out/ab
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab
run() { tag=$1; shift; (env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_l_$tag.json 2>&1; }
run tuner SY_X=1
run halo117 STREAMYOLO_HALO_TILES=117
run halo118 STREAMYOLO_HALO_TILES=118
run halo117_118 STREAMYOLO_HALO_TILES=117,118
run k1_121 STREAMYOLO_TILE_1X1K=121 STREAMYOLO_STREAM_1X1=0
run k1_121_122 STREAMYOLO_TILE_1X1K=121,122 STREAMYOLO_STREAM_1X1=0
run nostream STREAMYOLO_STREAM_1X1=0
run tuner2 SY_X=1
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],3), d.get('step_ms'), {k: v for k, v in d['roofline'].get('per_kind_ms').items() if k in ('conv','dgrad','wgrad')})" 2>&1 | cut -c1-300; done
