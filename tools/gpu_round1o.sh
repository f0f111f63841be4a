#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/pytest_gpu_w.log 2>&1
(timeout 600 python bench.py --workload train --model l --steps 10 --warmup 4 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_train_l_w.log 2>&1
(timeout 300 python bench.py --workload train --model s --steps 10 --warmup 4 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_train_s_w.log 2>&1
cat gpurun_out/pytest_gpu_w.log
for f in gpurun_out/bench_*_w.log; do echo $f; python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"].get("host_launch_ms_per_step"), d["roofline"]["per_kind_ms"])
PY
done
(timeout 600 python tools/profile_train.py 2>&1 | tail -75) > gpurun_out/profile_train_l_w.log 2>&1
head -45 gpurun_out/profile_train_l_w.log
