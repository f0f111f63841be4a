#!/bin/bash
# tools/halo_ablate.sh STAGE TILE SHAPE — in-kernel timelines (probe build) of one halo2 tile under the operand ablations of
# ConvArgs::ablate: which part of the main loop is memory-system time, which is LDS, which is the MFMA stream itself.
STAGE=$1; TILE=${2:-104}; SHAPE=${3:-10}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/$STAGE; mkdir -p $O
for ab in 0 1 2 3 32 64 96 128 131 163; do
    for m in dgrad; do
        echo "=== ablate $ab ($m)"
        STREAMYOLO_HIP_LIB=tools/probes/_build/libstreamyolo_probe.so python tools/kernel_timeline.py --kind conv --shape $SHAPE --tile $((TILE + 256 * ab)) --mode $m 2>&1 | grep -E "main loop done|lifetime|event-timed|first operands landed   "
    done
done > $O/halo_ablate_${TILE}_s${SHAPE}.txt 2>&1
cat $O/halo_ablate_${TILE}_s${SHAPE}.txt
