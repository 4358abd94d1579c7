"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass:
MI355X_MICROARCH.md 'rocprofv3 PMC slots').  Corrections as that guide's HBM section prescribes: the counters
are in KB; on gfx950 FETCH_SIZE reports half the bytes of wide (16 B / lane) streaming reads, so it is doubled;
WRITE_SIZE is exact for 16-B stores.

    python pmc_traffic.py FETCH.db WRITE.db OUT.json WORKLOAD BATCH [note]
"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(counter_value), sum(duration) from pmc_events where counter_name = ? group by name",
                       (counter,)).fetchall()
    return {r[0]: dict(launches=r[1], kb=r[2], ns=r[3]) for r in rows}


def short(name):
    n = name.split("(")[0]
    return n.replace("void ", "").replace("imcom::", "")


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k, f in fetch.items():
        if "imcom::" not in k:
            continue
        w = write.get(k, dict(launches=0, kb=0.0, ns=0))
        n = f["launches"]
        fb, wb = 2.0 * f["kb"] * 1024.0 / n, (w["kb"] * 1024.0 / w["launches"] if w["launches"] else 0.0)
        out[short(k)] = dict(launches=n, fetch_kb_raw_per_launch=f["kb"] / n, write_kb_per_launch=(w["kb"] / w["launches"] if w["launches"] else 0.0),
                             traffic_bytes_per_launch=fb + wb, avg_us_under_pmc=f["ns"] / n / 1e3)
    from pyimcom_amd._lib import source_sha16

    doc = dict(workload=sys.argv[4], batch=int(sys.argv[5]), csrc_sha16=source_sha16(), note=sys.argv[6] if len(sys.argv) > 6 else "", correction="traffic = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024 (gfx950, wide reads)",
               kernels=out)
    json.dump(doc, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"])[:12]:
        print(f"{k:32s} n={v['launches']:5d} traffic/launch {v['traffic_bytes_per_launch'] / 1e9:8.3f} GB  ({v['avg_us_under_pmc']:9.1f} us)")


if __name__ == "__main__":
    main()
