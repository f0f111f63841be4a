#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
(timeout 600 python tools/conv_probe.py --shapes 5 --tiles 102 --reps 5 --chain 20 2>&1 | tail -7) > gpurun_out/conv_probe_2g.log 2>&1
cat gpurun_out/conv_probe_2g.log
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -3) > gpurun_out/pytest_gpu_2g.log 2>&1
(timeout 600 python bench.py --workload train --model l --steps 10 --warmup 4 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_train_l_2g.log 2>&1
(timeout 300 python bench.py --workload infer --model l --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_infer_l_2g.log 2>&1
(timeout 300 python bench.py --workload train --model s --steps 10 --warmup 4 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_train_s_2g.log 2>&1
cat gpurun_out/pytest_gpu_2g.log
for f in gpurun_out/bench_*_2g.log; do echo $f; python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["per_kind_ms"], d["roofline"].get("frac"))
PY
done
