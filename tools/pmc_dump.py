"""Per-kernel totals of every counter in a rocprofv3 --pmc results db:  python tools/pmc_dump.py RESULTS.db [name-substring]"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
sub = sys.argv[2] if len(sys.argv) > 2 else "imcom::"
rows = cur.execute("select name, counter_name, count(distinct dispatch_id), sum(counter_value), sum(duration)/count(*) from pmc_events group by name, counter_name").fetchall()
d = {}
for n, c, k, v, t in rows:
    if sub not in n:
        continue
    sn = n.split("(")[0].replace("void ", "").replace("imcom::", "")
    d.setdefault(sn, {"launches": k, "avg_us": t / 1e3})[c] = v / k
for sn, v in d.items():
    print(sn, {k: (round(x, 1) if isinstance(x, float) else x) for k, x in v.items()})
