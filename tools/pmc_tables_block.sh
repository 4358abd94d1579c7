#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the table kernels for a block-like request (25 groups, 3267 tables): tools/pmc_tables_block.sh
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export PYTHONPATH=$ROOT
O=$ROOT/gpurun_out/tabpmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d $O/$c -o r -- python3 $ROOT/tools/check_profscope.py > $O/$c.log 2>&1 || echo "$c failed"
  python3 $ROOT/tools/pmc_dump.py $(find $O/$c -name "*.db" | head -1) inv_
done
