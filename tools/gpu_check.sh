#!/bin/bash
# quick GPU regression: full gpu suite + the two headline benches
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3) > gpurun_out/check_pytest.log 2>&1
cat gpurun_out/check_pytest.log
for w in "train l" "train s" "infer l"; do
set -- $w
timeout 600 python bench.py --workload $1 --model $2 --steps 10 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$w', round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['per_kind_ms'])"
done
