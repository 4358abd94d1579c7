#!/bin/bash
# L1 / TA counters of the builders (GPU box, repo root): tools/pmc_build_a.sh ["counter set" ...]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export PYTHONPATH=$ROOT
O=$ROOT/gpurun_out/apmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ $# -eq 0 ]; then set -- "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum"; fi
i=0
for set in "$@"; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o r -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-block --no-configs > /dev/null 2> $O/p$i.err || { echo "pass $i failed"; grep -m2 -i "error\|exceeds\|invalid\|not found" $O/p$i.err | cut -c1-300; continue; }
  python3 $ROOT/tools/pmc_dump.py $(find $O/p$i -name "*.db" | head -1) build_ | cut -c1-600
done
