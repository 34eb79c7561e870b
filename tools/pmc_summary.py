#!/usr/bin/env python3
"""Per-kernel means of the counters collected by tools/pmc_collect.sh:
   python tools/pmc_summary.py gpurun_out/pmc_<tag>_*/  [kernel-substring]   -> JSON on stdout
(launches of the warm-up are included; all launches of the bench batch are identical)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

dirs = [d for d in sys.argv[1:] if os.path.isdir(d)]
pat = [a for a in sys.argv[1:] if not os.path.isdir(a)]
acc = defaultdict(lambda: defaultdict(list))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        with open(f) as fh:
            for row in csv.DictReader(fh):
                key = (row["Dispatch_Id"], row["Counter_Name"])
                per_dispatch[key] += float(row["Counter_Value"])
                names[row["Dispatch_Id"]] = row["Kernel_Name"].split("(")[0]
        for (disp, ctr), v in per_dispatch.items():
            acc[names[disp]][ctr].append(v)
out = {}
for k, ctrs in acc.items():
    if pat and not any(p in k for p in pat):
        continue
    out[k] = {c: sum(v) / len(v) for c, v in sorted(ctrs.items())}
    out[k]["_launches"] = max(len(v) for v in ctrs.values())
print(json.dumps(out, indent=1))
