#!/bin/bash
# A/B on one box: the coaddition inside the backward launches (default) against the stand-alone epilogue (IMCOM_EPILOGUE_UNFUSED=1),
# headline + block leg, alternating (GPU box, repo root)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/gpurun_out/ab_fused; mkdir -p $O
cd $ROOT
for v in fused unfused fused unfused; do
  if [ $v = unfused ]; then export IMCOM_EPILOGUE_UNFUSED=1; else unset IMCOM_EPILOGUE_UNFUSED; fi
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --block-reps 1 > $O/bench_$v.json 2> $O/bench_$v.err || { echo "bench $v failed"; tail -3 $O/bench_$v.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$O/bench_$v.json").read().strip().splitlines()[-1])
st = d["stage_ms_per_step"]
print("$v", round(d["value"], 1), "stamps/s", round(d["ms_per_step"], 2), "ms/step; solve", round(st["solve_gemm"], 2), "epilogue", round(st["epilogue"], 3), "frac", round(d["roofline"]["frac"], 4), "| block", round(d["block"]["value"], 1), round(d["block"]["ms_per_block"], 1))
PY
done
