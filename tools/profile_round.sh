# Round profile: bench JSON, rocprofv3 kernel trace + stats, and three separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ busy/stall).
# Run on the GPU box from the repo root:  bash tools/profile_round.sh LABEL [bench.py args...] ; results under gpurun_out/prof/
# LABEL names the build in the emitted JSON (e.g. "r02_v3"); extra arguments go to every bench.py invocation.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ -f "$R/bench.py" ] || { echo "profile_round.sh: $R is not the repo root" >&2; exit 2; }
LABEL=${1:?usage: profile_round.sh LABEL [bench.py args]}
shift
ARGS="$*"
O=$R/gpurun_out/prof
rm -rf "$O"; mkdir -p "$O"
python "$R/bench.py" $ARGS > "$O/bench.json" 2> "$O/bench.err"
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- python3 "$R/bench.py" --no-cpu-baseline --no-configs $ARGS > "$O/kt_bench.json" 2> "$O/kt.err"
# the headline loop alone (no seams, no block leg: their single-stamp launches pull the per-kernel averages down): the averages of
# solve_fwd_kernel / solve_bwd_kernel here are what roofline.avg_launch_ms of the same run reports from its own HIP events
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kth" -- python3 "$R/bench.py" --no-cpu-baseline --no-block --no-configs $ARGS > "$O/kt_headline_bench.json" 2> "$O/kth.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$O/fetch" -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-block --no-configs $ARGS > /dev/null 2> "$O/fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$O/write" -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-block --no-configs $ARGS > /dev/null 2> "$O/write.err"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d "$O/sq" -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-block --no-configs $ARGS > /dev/null 2> "$O/sq.err"
cd "$R"
F=$(find gpurun_out/prof/fetch -name '*.db' | head -1); W=$(find gpurun_out/prof/write -name '*.db' | head -1); S=$(find gpurun_out/prof/sq -name '*.db' | head -1)
python tools/pmc_traffic.py "$F" "$W" gpurun_out/prof/pmc_traffic.json cfg2 256 "$LABEL: bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-block --no-configs $ARGS"
python tools/pmc_sq_summary.py "$S" gpurun_out/prof/pmc_sq_summary.txt "$LABEL: python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-block --no-configs $ARGS (2 steps in total)"
find gpurun_out/prof/kt gpurun_out/prof/kth -name '*stats*.csv' | head -5; rm -rf gpurun_out/prof/fetch gpurun_out/prof/write gpurun_out/prof/sq
cat gpurun_out/prof/bench.json; cat gpurun_out/prof/kt_bench.json
