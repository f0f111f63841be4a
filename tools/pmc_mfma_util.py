#!/usr/bin/env python3
"""MFMA utilisation per kernel family over the LAST training step of a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
pass (bench.py --workload train): util = MFMA-busy cycles summed over the chip / (kernel duration x clock x 1024 SIMDs), at the
2.4 GHz peak clock — a LOWER bound on the pipe occupancy, profiled passes run at 1.9-2.0 GHz (MI355X_MICROARCH.md, DVFS).
Usage: pmc_mfma_util.py <rocprof output dir> [clock GHz]"""
import csv
import glob
import re
import sys
from collections import defaultdict

d = sys.argv[1]
SIMDS = 256 * 4
CLOCK = float(sys.argv[2]) if len(sys.argv) > 2 else 2.4
disp = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        e = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"])})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
starts = sorted(v["s"] for v in disp.values() if "pack_weights_kernel" in v["name"])
t0 = starts[-1] if starts else 0
fam = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for v in disp.values():
    if v["s"] < t0:
        continue
    name = re.sub(r"^void ", "", v["name"].replace("(anonymous namespace)::", "")).split("(")[0]
    name = re.sub(r"^sy_conv::", "", name)
    e = fam[name]
    e[0] += 1; e[1] += v["e"] - v["s"]; e[2] += v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); e[3] += v.get("GRBM_GUI_ACTIVE", 0.0)
mf = {k: e for k, e in fam.items() if e[2] > 0}
print("%-56s %6s %9s %9s" % ("kernel (last step, MFMA kernels)", "calls", "ms", "MFMA util"))
for k, (n, dur, b, act) in sorted(mf.items(), key=lambda kv: -kv[1][1]):
    print("%-56s %6d %9.3f %8.1f%%" % (k[:56], n, dur / 1e6, 100.0 * b / (dur * CLOCK * SIMDS)))
td = sum(e[1] for e in mf.values()); tb = sum(e[2] for e in mf.values()); ta = sum(e[3] for e in mf.values())
print("%-56s %6d %9.3f %8.1f%%" % ("all MFMA kernels", sum(e[0] for e in mf.values()), td / 1e6, 100.0 * tb / (td * CLOCK * SIMDS)))
ad = sum(e[1] for e in fam.values()); aa = sum(e[3] for e in fam.values())
print("%-56s %6d %9.3f %8.1f%%   (MFMA-busy over every kernel of the step)" % ("whole step", sum(e[0] for e in fam.values()), ad / 1e6, 100.0 * tb / (ad * CLOCK * SIMDS)))
