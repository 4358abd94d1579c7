#!/bin/bash
# timing + L1 access counts of tools/bin/l1_access_probe (GPU box, repo root)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/gpurun_out/l1probe; rm -rf $O; mkdir -p $O
$ROOT/tools/bin/l1_access_probe | tee $O/timing.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum -d $O/p1 -o r -- $ROOT/tools/bin/l1_access_probe > /dev/null 2> $O/p1.err || { echo "pmc pass failed"; tail -3 $O/p1.err; }
python3 - $(find $O/p1 -name "*.db" | head -1) <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, counter_name, count(distinct dispatch_id), sum(counter_value) from pmc_events group by name, counter_name").fetchall()
ninstr = 256 * 3 * 40 * 4 * 64
for n, c, k, v in sorted(rows):
    if "probe_kernel" in n:
        print(n.split("(")[0][-24:], c, f"{v / k / ninstr:8.2f} per wave-instruction")
PY
