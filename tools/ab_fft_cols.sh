#!/bin/bash
# The column transform's workgroup shape, alternating on one box (GPU box, repo root): tools/ab_fft_cols.sh
#   default                  8 waves in one workgroup per CU, stage tables in LDS
#   IMCOM_FFT_COLS_SPLIT=1   one group of four waves per workgroup, as many workgroups per CU as the LDS holds
#   (a third shape, tried and removed: stage tables read from global memory so that the LDS holds twelve lines = three groups of four
#    waves: 2.68 us per table against 2.25, 2.57 as three independent workgroups per CU -- the 23 extra 16-byte loads per line cost more
#    than the third group brings)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for v in 0 1; do
    export IMCOM_FFT_COLS_SPLIT=$v
    echo -n "split $v: "; python3 $ROOT/tools/bench_fft_waves.py 60 12 2>&1 | grep "us/table"
  done
done
