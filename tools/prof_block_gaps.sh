#!/bin/bash
# kernel trace of the grouped block leg with the idle gaps listed (GPU box, repo root): tools/prof_block_gaps.sh LABEL [last_ms=420]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export PYTHONPATH=$ROOT
mkdir -p $ROOT/gpurun_out/blockprof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/blockprof -o $1 -- python3 $ROOT/tools/block_leg_only.py 2 > $ROOT/gpurun_out/blockprof/$1.log 2>&1
cd $ROOT
tail -2 gpurun_out/blockprof/$1.log
python tools/kernel_gaps.py gpurun_out/blockprof/$1_results.db 60 ${2:-420} > gpurun_out/blockprof/$1_gaps.txt
tail -60 gpurun_out/blockprof/$1_gaps.txt
