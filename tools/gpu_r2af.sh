#!/bin/bash
# round-2 stage af: K-chunked 1x1 tile kernel for Cin 1024 / 2048: tests, probes, A/B bench vs previous build
mkdir -p gpurun_out/af
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/af
PREV=$GRAFT_REPO_ROOT/streamyolo_amd/lib/ab_prev.so
(timeout 900 python -m pytest tests/test_kernels_conv.py -m gpu -q -x -k "conv1x1_tile" 2>&1 | tail -2) > $O/pytest_1x1.log 2>&1
tail -1 $O/pytest_1x1.log
for m in stats dgrad fwd; do
  echo "== mode $m" >> $O/conv_probe_bigk.txt
  (timeout 300 python tools/conv_probe.py --mode $m --shapes 12,21,22,23,17 --tiles 86,102,83,99,121 --reps 7 --chain 4 2>&1 | grep -v amdgpu.ids) >> $O/conv_probe_bigk.txt 2>&1
done
cat $O/conv_probe_bigk.txt
run() { tag=$1; shift; (env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $EXTRA 2>&1 | tail -1) > $O/bench_$tag.json 2>&1; }
run train_l_new SY_X=1
run train_l_prev STREAMYOLO_HIP_LIB=$PREV
EXTRA="--workload infer --model l" run infer_l_new SY_X=1
EXTRA="--workload infer --model l" run infer_l_prev STREAMYOLO_HIP_LIB=$PREV
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],3), d.get('step_ms'), {k: v for k, v in d['roofline'].get('per_kind_ms').items() if k in ('conv','dgrad','wgrad')})" 2>&1 | cut -c1-300; done
