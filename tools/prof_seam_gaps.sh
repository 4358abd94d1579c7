#!/bin/bash
# GPU idle gaps of the Block seam on a block of several passes (kernel trace of the last, warm call): tools/prof_seam_gaps.sh [n1P=48]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export PYTHONPATH=$ROOT
O=$ROOT/gpurun_out/seamgaps; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/t -o r -- python3 $ROOT/tools/profile_refblock32.py ${1:-48} 16 > $O/run.log 2>&1 || { echo failed; tail -3 $O/run.log; exit 1; }
grep "^call" $O/run.log
cd $ROOT
python3 tools/kernel_gaps.py $(find $O/t -name '*.db' | head -1) 3000 2300
