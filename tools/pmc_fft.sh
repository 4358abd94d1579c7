#!/bin/bash
# PMC passes over the table path (GPU box, repo root): tools/pmc_fft.sh ["counter set" ...]
# One derived counter (FETCH_SIZE, WRITE_SIZE) per pass: two together exceed what the hardware collects and rocprofv3 aborts.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export PYTHONPATH=$ROOT
O=$ROOT/gpurun_out/fftpmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ $# -eq 0 ]; then set -- "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; fi
i=0
for set in "$@"; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o r -- python3 $ROOT/tools/bench_fft_lines.py 48 8 12 1 > /dev/null 2> $O/p$i.err || { echo "pass $i failed"; grep -m1 -i "error code\|exceeds" $O/p$i.err; continue; }
  python3 $ROOT/tools/pmc_dump.py $(find $O/p$i -name "*.db" | head -1) _kernel
done
