import sqlite3,sys
db=sqlite3.connect(sys.argv[1]); cur=db.cursor()
rows=cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels group by name order by 3 desc limit %d"%int(sys.argv[2])).fetchall()
for r in rows: print("%-80s n=%6d total %9.2f ms avg %9.2f us"%(r[0][:80],r[1],r[2],r[3]))
