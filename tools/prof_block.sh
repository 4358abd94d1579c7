#!/bin/bash
# kernel trace of one grouped block (GPU box, repo root): tools/prof_block.sh LABEL [n1P] [batch]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export PYTHONPATH=$ROOT
mkdir -p $ROOT/gpurun_out/blockprof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/blockprof -o $1 -- python3 $ROOT/tools/bench_block.py ${2:-16} ${3:-128} > $ROOT/gpurun_out/blockprof/$1.log 2>&1
cd $ROOT
tail -3 gpurun_out/blockprof/$1.log
for k in inv_cols inv_rows fwd_rows fwd_cols; do python tools/kernel_seq.py gpurun_out/blockprof/$1_results.db $k 12; done
