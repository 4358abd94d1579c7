#!/bin/bash
# Matrix-pipe utilisation and wait shares of the Eigen path's kernels (cfg-3, batch 32, one stream): one --pmc pass of the SQ counters over
# tools/bench_eigen.py, reduced by tools/pmc_sq_summary.py (GPU box, repo root): tools/pmc_sq_eigen.sh [batch=32]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=${1:-32}
export PYTHONPATH=$ROOT IMCOM_EIGEN_SPLIT=1
O=$ROOT/gpurun_out/pmc_sq_eigen; rm -rf $O/sq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $O/sq -- python3 $ROOT/tools/bench_eigen.py cfg3 $B > $O/run.log 2>&1 || { echo "pass failed"; tail -3 $O/run.log; exit 1; }
cd $ROOT
S=$(find $O/sq -name '*.db' | head -1)
python tools/pmc_sq_summary.py "$S" $O/pmc_sq_eigen_b$B.txt "cfg-3 Eigen path, batch $B, one stream: python3 tools/bench_eigen.py cfg3 $B (3 steps)"
rm -rf $O/sq
head -16 $O/pmc_sq_eigen_b$B.txt
