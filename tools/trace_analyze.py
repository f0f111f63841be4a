#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 --kernel-trace CSV: for the last full training step (delimited by
pack_weights_kernel launches) report wall time, union-busy time, per-queue busy time, idle gaps and the
per-kernel-family totals.  Usage: python tools/trace_analyze.py <kernel_trace.csv> [--json OUT.json]
--json: the last step's per-kernel totals as a file bench.py reads (roofline.dominant.in_step_frac), stamped
with the kernel-source key of the tree it was measured on."""
import csv
import json
import os
import re
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
marks = [i for i, r in enumerate(rows) if "pack_weights_kernel" in r[2]]
if len(marks) < 2:
    print("no step delimiters found"); sys.exit(0)
a, b = marks[-2], marks[-1]
step = rows[a:b]
t0, t1 = step[0][0], max(r[1] for r in step)
print("kernels in step: %d   wall %.3f ms" % (len(step), (t1 - t0) / 1e6))
# union busy
ev = sorted((s, e) for s, e, _, _ in step)
busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
gaps = []
for s, e in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("union busy %.3f ms, idle %.3f ms in %d gaps (median gap %.1f us)" % (busy / 1e6, sum(gaps) / 1e6, len(gaps),
                                                                        sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0))
print("sum of kernel durations %.3f ms" % (sum(e - s for s, e, _, _ in step) / 1e6))
perq = defaultdict(lambda: [0, 0])
for s, e, n, q in step:
    perq[q][0] += 1; perq[q][1] += e - s
for q, (n, d) in perq.items():
    print("queue %s: %d kernels, %.3f ms busy" % (q, n, d / 1e6))
fam = defaultdict(lambda: [0, 0])
for s, e, n, q in step:
    m = re.search(r"(\w+_kernel|\w+)(<[^>]*>)?\(", n)
    key = re.sub(r"^void ", "", n.replace("(anonymous namespace)::", "").split("(")[0])[:70]
    fam[key][0] += 1; fam[key][1] += e - s
for k, (n, d) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:28]:
    print("%9.3f ms %5d  %s" % (d / 1e6, n, k))
MFMA_NAMES = ("conv_igemm_kernel", "conv3x3_halo", "conv1x1_tile_kernel", "conv_wgrad", "wgrad_fold")
mfma = sum(d for k, (n, d) in fam.items() if any(m in k for m in MFMA_NAMES))
print("MFMA kernels (conv_igemm + conv3x3_halo + conv1x1_tile + conv_wgrad + wgrad_fold) in the last step: %.3f ms  (bench.py roofline.kernel_ms_per_step measures the same set with HIP events)" % (mfma / 1e6))

if "--json" in sys.argv:
    out = sys.argv[sys.argv.index("--json") + 1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from streamyolo_amd import _lib
    try:
        commit = open(os.path.join(root, "tools", ".head_commit")).read().strip()
    except OSError:
        commit = None
    halo = sum(d for k, (n, d) in fam.items() if "conv3x3_halo" in k)
    rec = {"kernel_source_key": _lib.kernel_source_key(), "commit": commit,
           "what": "rocprofv3 --kernel-trace of bench.py (train l, 8 pairs, bf16): the LAST taped step between two pack_weights launches; "
                   "the trace serialises the streams on this stack, so these are per-launch durations of the step's own launch list",
           "launches_in_step": len(step), "sum_kernel_ms": sum(e - s for s, e, _, _ in step) / 1e6, "mfma_kernels_ms": mfma / 1e6,
           "conv3x3_halo_ms": halo / 1e6,
           "families_ms": {k: [n, round(d / 1e6, 4)] for k, (n, d) in sorted(fam.items(), key=lambda kv: -kv[1][1])}}
    with open(out, "w") as fh:
        json.dump(rec, fh, indent=1)
