#!/bin/bash
# quick GPU regression: full gpu suite + the headline benches (+ streaming with and without uint8 ingest)
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5) > gpurun_out/check_pytest.log 2>&1
cat gpurun_out/check_pytest.log
for w in "train l" "infer l"; do
set -- $w
timeout 600 python bench.py --workload $1 --model $2 --steps 10 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$w', round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['per_kind_ms'])" 2>&1 | tee -a gpurun_out/check_bench.log
done
for u in 0 1; do
timeout 300 python bench.py --workload stream --model l --dtype fp16 --steps 50 --warmup 10 --no-cpu-baseline --u8-input $u 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('stream u8=$u', round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['per_kind_ms'])" 2>&1 | tee -a gpurun_out/check_bench.log
done
