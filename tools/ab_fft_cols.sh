#!/bin/bash
# The column transform's workgroup shape: IMCOM_FFT_COLS_SPLIT=1 (one group of four waves per workgroup, two workgroups per CU) against the
# default (8 waves in one workgroup per CU), alternating on one box: tools/ab_fft_cols.sh   (GPU box, repo root)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export PYTHONPATH=$ROOT IMCOM_HIP_LIB=${IMCOM_HIP_LIB:-$ROOT/pyimcom_amd/lib/libimcom_hip.so}
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for v in 1 0; do
    export IMCOM_FFT_COLS_SPLIT=$v
    echo -n "cols split $v: "; python3 $ROOT/tools/bench_fft_waves.py 60 12 2>&1 | grep "us/table"
  done
done
unset IMCOM_FFT_COLS_SPLIT
python3 $ROOT/tools/bench_fft_lines.py 48 8 12 3 2>&1 | tail -2
