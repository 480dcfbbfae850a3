import sqlite3, sys
for name in sys.argv[1:]:
    con = sqlite3.connect(name); cur = con.cursor()
    rows = cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(name, "total kernel ms %.2f" % (tot/1e3))
    for r in rows[:12]:
        nm = r[0].split('(')[0].replace('void ctv::','')[:40]
        print(f"  {nm:40s} n={r[1]:5d} total={r[2]/1e3:9.3f} ms avg={r[3]:9.2f} us  {100*r[2]/tot:5.1f}%")
