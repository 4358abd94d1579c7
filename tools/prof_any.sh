#!/bin/bash
# kernel stats of any script under rocprofv3 (GPU box, repo root): tools/prof_any.sh LABEL script.py [args]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export PYTHONPATH=$ROOT
L=$1; shift
mkdir -p $ROOT/gpurun_out/anyprof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/anyprof/$L -- python3 $ROOT/$@ > $ROOT/gpurun_out/anyprof/$L.log 2>&1
cd $ROOT
F=$(find gpurun_out/anyprof/$L -name '*kernel_stats.csv' | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:30]:
    print(f"{r['Name'][:70]:70s} n={int(r['Calls']):5d} avg={float(r['AverageNs'])/1e3:9.1f} us  tot={float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
