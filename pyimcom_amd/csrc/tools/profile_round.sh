# Round profile: bench JSON, rocprofv3 kernel trace + stats, and three separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ busy/stall).
# Run on the GPU box from the repo root:  bash pyimcom_amd/csrc/tools/profile_round.sh ; results under gpurun_out/prof/
set -e
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof; mkdir -p $R/gpurun_out/prof
python bench.py > $R/gpurun_out/prof/bench.json 2> $R/gpurun_out/prof/bench.err
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/kt -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof/kt_bench.json 2> $R/gpurun_out/prof/kt.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof/fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof/write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof/write.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/prof/sq -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof/sq.err
cd $R
rm -rf gpurun_out/prof_old; F=$(find gpurun_out/prof/fetch -name '*.db' | head -1); W=$(find gpurun_out/prof/write -name '*.db' | head -1); S=$(find gpurun_out/prof/sq -name '*.db' | head -1)
python pyimcom_amd/csrc/tools/pmc_traffic.py $F $W gpurun_out/prof/pmc_traffic.json cfg2 256 "round 1 final build (v7)"
python pyimcom_amd/csrc/tools/pmc_sq_summary.py $S gpurun_out/prof/pmc_sq_summary.txt "python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline (cfg2, batch 256, 2 steps in total; final round-1 build)"
find gpurun_out/prof/kt -name '*stats*.csv' | head; rm -rf gpurun_out/prof/fetch gpurun_out/prof/write gpurun_out/prof/sq
cat gpurun_out/prof/bench.json; cat gpurun_out/prof/kt_bench.json
