#!/bin/bash
# L1 / L2 counters of the A builder inside the block leg (four PSF groups per stamp) (GPU box, repo root)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export PYTHONPATH=$ROOT
O=$ROOT/gpurun_out/apmc_block; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "FETCH_SIZE_sum" "WRITE_SIZE_sum"; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o r -- python3 $ROOT/tools/block_leg_only.py 1 > /dev/null 2> $O/p$i.err || { echo "pass $i failed"; grep -m2 -i "error\|exceeds\|invalid\|not found" $O/p$i.err | cut -c1-300; continue; }
  python3 $ROOT/tools/pmc_dump.py $(find $O/p$i -name "*.db" | head -1) build_A | cut -c1-600
done
