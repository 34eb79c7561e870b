#!/usr/bin/env python3
"""Per-kernel means of the counters collected by tools/profile_round.sh / tools/pmc_collect.sh:
   python tools/pmc_summary.py gpurun_out/pmc_<tag>_*/  [kernel-substring]   -> JSON on stdout
Launches are grouped by (kernel, grid size): a kernel that serves several query types in one run (the fill kernel
expands the carriers of the type-6 batch and of the type-4 leg) gets one entry per launch shape, keyed
"<kernel> [grid <n>]"; within a group all launches of the bench are identical.  `_avg_us` is the mean launch duration
under the counters (Start/End timestamps of the same rows)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

dirs = [d for d in sys.argv[1:] if os.path.isdir(d)]
pat = [a for a in sys.argv[1:] if not os.path.isdir(a)]
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        with open(f) as fh:
            for row in csv.DictReader(fh):
                key = (row["Dispatch_Id"], row["Counter_Name"])
                per_dispatch[key] += float(row["Counter_Value"])
                if row["Dispatch_Id"] not in names:
                    names[row["Dispatch_Id"]] = f'{row["Kernel_Name"].split("(")[0]} [grid {row["Grid_Size"]}]'
                    dur[names[row["Dispatch_Id"]]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
        for (disp, ctr), v in per_dispatch.items():
            acc[names[disp]][ctr].append(v)
out = {}
for k, ctrs in acc.items():
    if pat and not any(p in k for p in pat):
        continue
    out[k] = {c: sum(v) / len(v) for c, v in sorted(ctrs.items())}
    out[k]["_launches"] = max(len(v) for v in ctrs.values())
    out[k]["_avg_us"] = sum(dur[k]) / len(dur[k])
print(json.dumps(out, indent=1))
