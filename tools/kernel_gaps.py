"""Idle gaps of the GPU between consecutive kernel dispatches, from a rocprofv3 --kernel-trace results db:
   python tools/kernel_gaps.py RESULTS.db [min_gap_us=100] [last_ms=0: whole trace, else only the last N ms]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
last = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
if last > 0:
    t1 = rows[-1][2]
    rows = [r for r in rows if r[1] >= t1 - last * 1e6]
short = lambda n: n.split("(")[0].replace("imcom::", "").replace("void ", "")[:40]
tot, busy_end = 0.0, rows[0][2]
print(f"{len(rows)} dispatches over {(rows[-1][2] - rows[0][1]) / 1e6:.2f} ms")
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    gap = (s1 - busy_end) / 1e3
    if gap > 0:
        tot += gap
        if gap >= thr:
            print(f"  {gap:8.0f} us idle after {short(n0):40s} before {short(n1):40s} at {(s1 - rows[0][1]) / 1e6:8.2f} ms")
    busy_end = max(busy_end, e1)
print(f"total idle {tot / 1e3:.2f} ms")
