#!/bin/bash
# LDS / issue counters of the tile engine's k loop as a plain product (GPU box, repo root): tools/pmc_gemm_probe.sh ["counter set" ...]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export PYTHONPATH=$ROOT
O=$ROOT/gpurun_out/gemmpmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ $# -eq 0 ]; then set -- "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS"; fi
i=0
for set in "$@"; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o r -- python3 $ROOT/tools/probe_layouts.py 2304 8 > /dev/null 2> $O/p$i.err || { echo "pass $i failed"; grep -m2 -i "error\|exceeds\|invalid" $O/p$i.err | cut -c1-200; continue; }
  python3 $ROOT/tools/pmc_dump.py $(find $O/p$i -name "*.db" | head -1) _kernel
done
