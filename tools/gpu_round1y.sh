#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3) > gpurun_out/pytest_gpu_y.log 2>&1
(timeout 600 python bench.py --workload train --model l --steps 10 --warmup 4 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_train_l_y.log 2>&1
(timeout 300 python bench.py --workload train --model s --steps 10 --warmup 4 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_train_s_y.log 2>&1
(timeout 300 python bench.py --workload infer --model l --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_infer_l_y.log 2>&1
cat gpurun_out/pytest_gpu_y.log
for f in gpurun_out/bench_*_y.log; do echo $f; python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"].get("host_launch_ms_per_step"), d["roofline"]["per_kind_ms"])
PY
done
(timeout 900 python tools/pmc_traffic.py --out gpurun_out/traffic_train_l.json -- --workload train --model l 2>&1 | tail -14) > gpurun_out/traffic_train_l.log 2>&1
cat gpurun_out/traffic_train_l.log
