#!/bin/bash
# A/B of the band reduction's lazy super-panel width (IMCOM_BTPL = 64 | 128: trailing update every 64 / 128 reflectors) on one box
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/gpurun_out/ab_btpl; mkdir -p $O
cd $ROOT
export PYTHONPATH=$ROOT IMCOM_EIGEN_SPLIT=1
for u in 64 128 64 128; do
  touch pyimcom_amd/csrc/band.hip
  make -s -C pyimcom_amd/csrc EXTRA="-DIMCOM_BTPL=$u" > $O/make_$u.log 2>&1 || { echo "build $u failed"; tail -5 $O/make_$u.log; exit 1; }
  if [ $u = 128 ]; then timeout -k 10 200 python -m pytest tests/test_gpu_band.py -m gpu -x -q > $O/pytest_$u.log 2>&1; echo "band tests BTPL=$u rc=$?"; fi
  for b in 32 256; do
    timeout -k 10 200 python tools/bench_eigen.py cfg3 $b > $O/eig_${u}_b$b.json 2> $O/eig_${u}_b$b.err || { echo "bench $u $b failed"; tail -3 $O/eig_${u}_b$b.err; continue; }
    python -c "
import json; d=json.load(open('$O/eig_${u}_b$b.json')); print('BTPL=$u batch $b:', round(d['ms_per_stamp'],3), 'ms per stamp;', d['stage_ms_per_step'])"
  done
done
touch pyimcom_amd/csrc/band.hip
