#!/bin/bash
# HBM traffic per launch of the Iterative kernel's launches (iter_default, batch 128: tools/bench_iter.py 128 1 0): FETCH_SIZE and WRITE_SIZE in
# separate --pmc passes, corrected as tools/pmc_traffic.py documents (GPU box, repo root) -> gpurun_out/pmc_iter/pmc_traffic_iter.json
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export PYTHONPATH=$ROOT
O=$ROOT/gpurun_out/pmc_iter; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -- python3 $ROOT/tools/bench_iter.py 128 1 0 > $O/fetch.log 2>&1 || { echo "fetch pass failed"; tail -3 $O/fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -- python3 $ROOT/tools/bench_iter.py 128 1 0 > $O/write.log 2>&1 || { echo "write pass failed"; tail -3 $O/write.log; exit 1; }
cd $ROOT
F=$(find $O/fetch -name '*.db' | head -1); W=$(find $O/write -name '*.db' | head -1)
python tools/pmc_traffic.py "$F" "$W" $O/pmc_traffic_iter.json iter_default 128 "r06: python3 tools/bench_iter.py 128 1 0 (one warm-up step + one timed step of 128 default-configuration stamps)"
rm -rf $O/fetch $O/write
