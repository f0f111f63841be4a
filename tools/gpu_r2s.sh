#!/bin/bash
# round-2 stage s: RETIRED timing experiment (the STREAMYOLO_EXP_SKIP_FINALIZE switch no longer exists; see profiles/r02/README.md: its result was an artefact)
mkdir -p gpurun_out/s
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/s
run() { tag=$1; shift; (env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_l_$tag.json 2>&1; }
run base SY_X=1
run nofinalize STREAMYOLO_EXP_SKIP_FINALIZE=1
run base2 SY_X=1
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],3), d.get('step_ms'))" 2>&1 | cut -c1-300; done
