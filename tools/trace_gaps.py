import sqlite3, glob, sys
db=sys.argv[1]
c=sqlite3.connect(db)
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kt=[t for t in tabs if 'kernel_dispatch' in t][0]
st=[t for t in tabs if 'info_kernel_symbol' in t][0]
rows=c.execute(f"select s.kernel_name, d.start, d.end, d.queue_id from {kt} d join {st} s on d.kernel_id=s.id order by d.start").fetchall()
idx=[i for i,r in enumerate(rows) if 'k_fill_sites2' in r[0]]
def short(n): return n.split('(')[0].replace('vsamd::','').replace('void ','').replace('_ZN5vsamd','')[:14]
for a,b in zip(idx[4:12], idx[5:13]):
    ra, rb = rows[a], rows[b]
    inter=[(short(r[0]), round((r[1]-ra[1])/1e3,1), round((r[2]-r[1])/1e3,1)) for r in rows[a+1:b] if 't6' in r[0]]
    print(f"fill dur {(ra[2]-ra[1])/1e3:7.1f} us | gap {(rb[1]-ra[2])/1e3:5.1f} | plan kernels (start after fill start, dur):", inter)
