"""The kernels of one type-4 batch in a rocprofv3 --kernel-trace database, in stream order: start (us after the batch's first kernel), duration."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if 'kernel_dispatch' in t][0]
st = [t for t in tabs if 'info_kernel_symbol' in t][0]
rows = c.execute(f"select s.kernel_name, d.start, d.end from {kt} d join {st} s on d.kernel_id=s.id order by d.start").fetchall()
key = sys.argv[2] if len(sys.argv) > 2 else 'k_sample_walk_coop'
idx = [i for i, r in enumerate(rows) if key in r[0]]
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0 = rows[a][1]
for r in rows[a - 6:b + 1]:
    print(f"{(r[1] - t0) / 1e3:8.1f} {(r[2] - r[1]) / 1e3:7.1f}  {r[0].split('(')[0][:60]}")
