#!/bin/bash
# round-2 stage z: one-pass top-k in the SimOTA assignment kernel, arena views in one call on the drop-in path: tests + A/B
mkdir -p gpurun_out/z
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/z
PREV=$GRAFT_REPO_ROOT/streamyolo_amd/lib/ab_prev.so
(timeout 1200 python -m pytest tests/test_kernels_loss.py tests/test_model_train.py tests/test_amp.py -m gpu -q -x 2>&1 | grep -vE 'RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids' | tail -3) > $O/pytest_loss_train.log 2>&1
tail -2 $O/pytest_loss_train.log
run() { tag=$1; shift; (env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $EXTRA 2>&1 | tail -1) > $O/bench_train_l_$tag.json 2>&1; }
run new SY_X=1
run prev STREAMYOLO_HIP_LIB=$PREV
EXTRA="--path dropin" run dropin_new SY_X=1
EXTRA="--batch 4" run b4_new SY_X=1
EXTRA="--batch 4" run b4_prev STREAMYOLO_HIP_LIB=$PREV
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(round(d['value'],1), round(d['ms_per_step'],3), d.get('step_ms'), d['config'].get('host_launch_ms_per_step'))" 2>&1 | cut -c1-300; done
(cd /tmp && rm -rf /tmp/prof_z && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_z -- python $GRAFT_REPO_ROOT/bench.py --workload train --model l --steps 5 --warmup 3 --no-cpu-baseline > /dev/null 2>&1)
python tools/trace_analyze.py $(ls /tmp/prof_z/*/*kernel_trace.csv | head -1) 2>&1 | grep -E "tal_|kernels in step|sum of"
