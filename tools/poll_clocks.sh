#!/bin/bash
# shader / memory clocks while a command runs (GPU box): tools/poll_clocks.sh "command"   -> gpurun_out/clocks.log
mkdir -p gpurun_out
( bash -c "$1" > gpurun_out/clocks_cmd.log 2>&1 ) &
PID=$!
sleep ${2:-8}
for i in $(seq 1 ${3:-12}); do
  kill -0 $PID 2>/dev/null || break
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|fclk\|socclk\|power" | tr -s ' ' | tr '\n' ';'
  echo
  sleep ${4:-0.5}
done | tee gpurun_out/clocks.log
wait $PID
tail -2 gpurun_out/clocks_cmd.log | cut -c1-200
