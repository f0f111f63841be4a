#!/bin/bash
# round-2 stage x: persistent 1x1 tile kernel (124 / 125): tests, probes, A/B bench (tuner with / without the candidates)
mkdir -p gpurun_out/x
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/x
(timeout 900 python -m pytest tests/test_kernels_conv.py -m gpu -q -x -k "conv1x1" 2>&1 | tail -2) > $O/pytest_1x1.log 2>&1
tail -1 $O/pytest_1x1.log
S1=2,4,5,7,9,16,18,19,20
for m in stats dgrad fwd; do
  echo "== mode $m" >> $O/conv_probe_ptile.txt
  (timeout 300 python tools/conv_probe.py --mode $m --shapes $S1 --tiles 86,120,121,122,124,125 --reps 7 --chain 4 2>&1 | grep -v amdgpu.ids) >> $O/conv_probe_ptile.txt 2>&1
done
for b in 256 1024; do
  echo "== stats, SY_PTILE_BLOCKS=$b" >> $O/conv_probe_ptile.txt
  (SY_PTILE_BLOCKS=$b timeout 300 python tools/conv_probe.py --mode stats --shapes $S1 --tiles 124,125 --reps 7 --chain 4 2>&1 | grep -v amdgpu.ids) >> $O/conv_probe_ptile.txt 2>&1
done
cat $O/conv_probe_ptile.txt
run() { tag=$1; shift; (env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_train_l_$tag.json 2>&1; }
run new SY_X=1
run no_ptile STREAMYOLO_TILE_1X1K=121,122,123
(timeout 300 python bench.py --workload infer --model l --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_infer_l.json 2>&1
(STREAMYOLO_TILE_1X1K=121,122,123 timeout 300 python bench.py --workload infer --model l --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_infer_l_no_ptile.json 2>&1
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],3), d.get('step_ms'), d['roofline'].get('per_kind_ms'))" 2>&1 | cut -c1-500; done
