"""Per-dispatch durations of one kernel, in launch order, from a rocprofv3 --kernel-trace results db:
   python tools/kernel_seq.py RESULTS.db NAME-SUBSTRING [count]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
sel = [(n, s, e) for n, s, e in rows if sys.argv[2] in n]
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 40
print(len(sel), "dispatches of", sys.argv[2])
print(" ".join(f"{(e - s) / 1e3:.0f}" for n, s, e in sel[-cnt:]))
