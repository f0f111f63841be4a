#!/bin/bash
# tools/gpu_ab_tree.sh STAGE N [BENCH ARGS] — same-box alternating A/B of THIS tree against another checkout of the repository in
# ./_base (a `git worktree` of the commit to compare with, built in place): N rounds of [this tree, _base], each a default bench.py
# line without extras.  Writes gpurun_out/STAGE/ab_{new,base}_K.json and a summary.
STAGE=$1; N=${2:-3}; shift; shift
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=$PWD/gpurun_out/$STAGE
mkdir -p "$O"
for k in $(seq 1 $N); do
    (timeout 900 python bench.py --extras 0 --no-cpu-baseline "$@" 2>$O/ab_new_$k.err | tail -1) > $O/ab_new_$k.json
    (cd _base && timeout 900 python bench.py --extras 0 --no-cpu-baseline "$@" 2>$O/ab_base_$k.err | tail -1) > $O/ab_base_$k.json
done
python - "$O" "$N" <<'PY'
import json, sys
o, n = sys.argv[1], int(sys.argv[2])
for tag in ("new", "base"):
    ms = []
    for k in range(1, n + 1):
        try:
            d = json.load(open("%s/ab_%s_%d.json" % (o, tag, k)))
            ms.append((d["ms_per_step"], d["config"].get("launches_per_step"), d["config"].get("host_issue_ms_per_step")))
        except Exception as e:
            ms.append((float("nan"), None, None))
    print(tag, " ".join("%.3f" % m[0] for m in ms), "ms/step; launches", ms[-1][1], "host issue", ms[-1][2])
PY
