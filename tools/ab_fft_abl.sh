#!/bin/bash
# Ablation of the line FFTs: libimcom_hip_abl<n>.so = the product library with psf_overlap.o compiled with -DIMCOM_FFT_ABL=n (fft_lines.h;
#   hipcc $CXXFLAGS -DIMCOM_FFT_ABL=n -c psf_overlap.hip -o build_abln/psf_overlap.o; hipcc -shared -o ../lib/libimcom_hip_abln.so build_abln/*.o): per-kernel times of a block-sized
# table request under rocprofv3 --kernel-trace --stats.   tools/ab_fft_abl.sh [n ...]      (GPU box, repo root)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export PYTHONPATH=$ROOT
O=$ROOT/gpurun_out/fftabl; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
[ $# -eq 0 ] && set -- 0 1 2 3 7 11 15
for n in "$@"; do
  export IMCOM_HIP_LIB=$ROOT/pyimcom_amd/lib/libimcom_hip_abl$n.so
  [ -f $IMCOM_HIP_LIB ] || { echo "no library for $n"; continue; }
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/v$n -- python3 $ROOT/tools/bench_fft_waves.py 60 12 > $O/v$n.log 2>&1 || { echo "variant $n failed"; tail -3 $O/v$n.log; continue; }
  F=$(find $O/v$n -name '*kernel_stats.csv' | head -1)
  python3 - "$F" $n <<'PY'
import csv, sys
rows = {r['Name'].split('(')[0].replace('void imcom::', ''): r for r in csv.DictReader(open(sys.argv[1]))}
out = []
for k, r in rows.items():
    if 'inv_' in k:
        out.append(f"{k.split('<')[0]} avg {float(r['AverageNs'])/1e3:8.1f} us (n={r['Calls']})")
print(f"ABL {int(sys.argv[2]):2d}: " + "   ".join(sorted(out)))
PY
  rm -rf $O/v$n
done
