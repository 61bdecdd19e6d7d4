#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per kernel name, mean counter value per dispatch (moai kernels only)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for f in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "moai::" not in k:
            continue
        k = k.replace("void moai::", "").split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(root, "p*", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "moai::" not in k:
            continue
        k = k.replace("void moai::", "").split("(")[0]
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {}
for k in sorted(acc):
    out[k] = {c: sum(v) / len(v) for c, v in sorted(acc[k].items())}
    if dur[k]:
        out[k]["_avg_us_profiled"] = sum(dur[k]) / len(dur[k])
    out[k]["_dispatches"] = max(len(v) for v in acc[k].values())
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
