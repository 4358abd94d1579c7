#!/bin/bash
# SQ / LDS / VMEM counter passes over a block-sized table request (tools/bench_fft_waves.py, 3600 tables, 12 waves):
#   tools/pmc_fft_block.sh ["counter set" ...]        (GPU box, repo root)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export PYTHONPATH=$ROOT
O=$ROOT/gpurun_out/fftpmc2; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ $# -eq 0 ]; then set -- "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_FLAT SQ_IFETCH SQ_INSTS_SALU" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; fi
i=0
for set in "$@"; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o r -- python3 $ROOT/tools/bench_fft_waves.py 60 12 > /dev/null 2> $O/p$i.err || { echo "pass $i failed"; grep -m1 -i "error code\|exceeds\|not found\|invalid" $O/p$i.err; continue; }
  python3 $ROOT/tools/pmc_dump.py $(find $O/p$i -name "*.db" | head -1) inv_
done
