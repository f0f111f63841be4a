#!/usr/bin/env python3
"""Mean counter values / durations per kernel name from a rocprofv3 --pmc ... --kernel-trace CSV directory."""
import csv
import glob
import re
import sys
from collections import defaultdict

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "conv_igemm"
cnt = defaultdict(lambda: defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            key = (re.sub(r"^.*?(\w+)<(.*)>.*", r"\1<\2>", r["Kernel_Name"]), r.get("Grid_Size", ""))
            cnt[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            key = (re.sub(r"^.*?(\w+)<(.*)>.*", r"\1<\2>", r["Kernel_Name"]), r.get("Grid_Size", r.get("Grid_Size_X", "")))
            dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for key in sorted(set(cnt) | set(dur)):
    c = {k: sum(v) / len(v) for k, v in cnt.get(key, {}).items()}
    dd = dur.get(key, [])
    print(key, "n=%d" % max(len(dd), max([len(v) for v in cnt.get(key, {}).values()] or [0])),
          "dur_us=%.1f" % (sum(dd) / len(dd) / 1e3) if dd else "", {k: round(v) for k, v in sorted(c.items())})
