#!/bin/bash
# kernel trace of the block leg with the idle gaps listed (GPU box, repo root): tools/prof_block_gaps.sh LABEL [min_gap_us=300]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export PYTHONPATH=$ROOT
mkdir -p $ROOT/gpurun_out/blockprof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/blockprof -o $1 -- python3 $ROOT/tools/block_leg_only.py 1 > $ROOT/gpurun_out/blockprof/$1.log 2>&1
cd $ROOT
tail -2 gpurun_out/blockprof/$1.log | cut -c1-1500
python tools/kernel_gaps.py gpurun_out/blockprof/$1_results.db ${2:-300} 2300 > gpurun_out/blockprof/$1_gaps.txt
tail -70 gpurun_out/blockprof/$1_gaps.txt
rm -f gpurun_out/blockprof/$1_results.db
