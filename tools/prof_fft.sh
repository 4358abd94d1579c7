#!/bin/bash
# kernel durations of the table path under rocprofv3 (run on the GPU box from the repo root): tools/prof_fft.sh LABEL
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export PYTHONPATH=$ROOT
mkdir -p $ROOT/gpurun_out/fftprof
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $ROOT/gpurun_out/fftprof -o $1 -- python3 $ROOT/tools/bench_fft_lines.py 48 8 12 3 > /dev/null 2>&1
cd $ROOT
for k in inv_cols inv_rows fwd_rows fwd_cols; do python tools/kernel_seq.py gpurun_out/fftprof/$1_results.db $k 4; done
