# per-launch durations of the headline step's Cholesky and solve launches
set -e
R=$(pwd); O=$R/gpurun_out/s2; mkdir -p $O; rm -rf $O/kt
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
rocprofv3 --kernel-trace -d $O/kt -- python3 $R/bench.py --no-cpu-baseline --no-block --no-configs --steps 2 --warmup 1 > $O/kt_bench.json 2> $O/kt.err
cd $R
DB=$(find $O/kt -name '*.db' | head -1)
for k in chol_update chol_trsm chol_diag solve_fwd solve_bwd; do python tools/kernel_seq.py $DB $k 18; done > $O/kernel_seq.txt
cat $O/kernel_seq.txt; rm -rf $O/kt
