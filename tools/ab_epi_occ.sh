#!/bin/bash
# Is the coaddition epilogue bound by the workgroups a CU holds?  IMCOM_EPI_LDS_PAD pads its LDS request: 0 = two workgroups per CU, 45000 = one.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
for pad in 0 45000 0 45000; do
  export IMCOM_EPI_LDS_PAD=$pad
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-block --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pad $pad:', round(d['ms_per_step'],2), 'ms/step; epilogue', round(d['stage_ms_per_step']['epilogue'],3))"
done
