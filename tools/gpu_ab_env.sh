#!/bin/bash
# tools/gpu_ab_env.sh STAGE N "ENV_A" "ENV_B" ... — same-box alternating A/B of environment-switch variants of THIS tree: N rounds over
# the variants (each a space-separated list of VAR=value, "-" for none), each a default bench.py line without extras.
# Writes gpurun_out/STAGE/env_<i>_<k>.json and a summary table.
STAGE=$1; N=$2; shift; shift
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=$PWD/gpurun_out/$STAGE
mkdir -p "$O"
printf '%s\n' "$@" > $O/variants.txt
for k in $(seq 1 $N); do
    i=0
    for v in "$@"; do
        [ "$v" = "-" ] && v=""
        (env $v timeout 900 python bench.py --extras 0 --no-cpu-baseline $BENCH_ARGS 2>$O/env_${i}_$k.err | tail -1) > $O/env_${i}_$k.json
        i=$((i + 1))
    done
done
python - "$O" "$N" "$#" <<'PY' | tee $O/summary.txt
import json, sys
o, n, nv = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
names = open(o + "/variants.txt").read().splitlines()
for i in range(nv):
    ms = []
    for k in range(1, n + 1):
        try:
            ms.append(json.load(open("%s/env_%d_%d.json" % (o, i, k)))["ms_per_step"])
        except Exception:
            ms.append(float("nan"))
    print("%-60s %s  ms/step" % (names[i], " ".join("%.3f" % m for m in ms)))
PY
