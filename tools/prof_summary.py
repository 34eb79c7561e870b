#!/usr/bin/env python3
"""Summarise a rocprofv3 results .db (kernel-trace --stats) as a text table:
   python tools/prof_summary.py gpurun_out/prof_x/x_results.db > profiles/rNN_x.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print(f"# rocprofv3 --kernel-trace --stats summary of {sys.argv[1]}")
if len(sys.argv) > 2:
    print("# command: " + " ".join(sys.argv[2:]))
print(f"{'kernel':<90} {'calls':>6} {'total_us':>14} {'avg_us':>12} {'pct':>7}")
for name, calls, tot, avg, pct in rows:
    print(f"{name[:90]:<90} {calls:>6} {tot:>14.3f} {avg:>12.3f} {pct:>7.2f}")
