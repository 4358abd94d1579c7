#!/bin/bash
# Per-launch durations of the band reduction's kernels along one reduction (cfg-3, one stream): tools/symv_profile.sh [batch=32]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=${1:-32}
export PYTHONPATH=$ROOT IMCOM_EIGEN_SPLIT=1 IMCOM_EIGEN_OVERLAP=0
O=$ROOT/gpurun_out/symvprof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/t -o r -- python3 $ROOT/tools/bench_eigen.py cfg3 $B > $O/run.log 2>&1 || { echo failed; tail -3 $O/run.log; exit 1; }
tail -1 $O/run.log
cd $ROOT
python3 - $(find $O/t -name '*.db' | head -1) $B <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
B = int(sys.argv[2])
for key in ("symv4_kernel", "band_step_reg_kernel", "band_apply_kernel"):
    sel = [(s, e) for n, s, e in rows if key in n]
    per = len(sel) // 3  # three timed steps
    last = sel[-per:]
    d = [(e - s) / 1e3 for s, e in last]
    print(key, "launches per reduction", per, "total ms", round(sum(d) / 1e3, 2))
    print("  every 40th:", " ".join(f"{x:.0f}" for x in d[::40]))
# gaps between consecutive kernels inside the last reduction
sel = [(n, s, e) for n, s, e in rows if any(k in n for k in ("symv4_kernel", "band_step", "band_apply", "syr2k"))]
per = len(sel) // 3
last = sel[-per:]
gaps = [(last[i + 1][1] - last[i][2]) / 1e3 for i in range(len(last) - 1)]
busy = sum((e - s) for n, s, e in last) / 1e6
print("reduction span ms", round((last[-1][2] - last[0][1]) / 1e6, 2), "kernel busy ms", round(busy, 2), "sum of gaps ms", round(sum(g for g in gaps if g > 0) / 1e3, 2), "median gap us", sorted(gaps)[len(gaps) // 2])
PY
