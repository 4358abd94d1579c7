#!/bin/bash
# Does the tables' column transform gain from a third wave per SIMD?  At nfft = 512 a line takes 8.7 KB of LDS and twelve waves fit
# (at 768: eight).  Library built from commit 6a9c2be (stage tables in LDS, 132 VGPRs: up to three waves per SIMD), IMCOM_FFT_WAVES = 12 / 8,
# per-kernel times of 60 x 60 tables under rocprofv3.       tools/ab_fft_cols_occ.sh LIB      (GPU box, repo root)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export PYTHONPATH=$ROOT IMCOM_HIP_LIB=${1:-$ROOT/pyimcom_amd/lib/libimcom_hip_ldstw.so}
O=$ROOT/gpurun_out/fftocc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in 12 8 12 8; do
  export IMCOM_FFT_WAVES=$w
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/w$w -- python3 $ROOT/tools/bench_fft_lines.py 32 8 60 3 > $O/w$w.log 2>&1 || { echo "failed"; tail -3 $O/w$w.log; continue; }
  F=$(find $O/w$w -name '*kernel_stats.csv' | head -1)
  python3 - "$F" $w <<'PY'
import csv, sys
rows = {r['Name'].split('(')[0].replace('void imcom::', ''): r for r in csv.DictReader(open(sys.argv[1]))}
print(f"waves {sys.argv[2]}: " + "   ".join(f"{k.split('<')[0]} {float(r['AverageNs'])/1e3:8.1f} us (n={r['Calls']})" for k, r in sorted(rows.items()) if 'inv_' in k))
PY
  rm -rf $O/w$w
done
