#!/bin/bash
# A/B of the coaddition epilogue's rows in flight per thread (IMCOM_EPI_U) on one box: headline bench per variant (GPU box, repo root)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/gpurun_out/ab_epi; mkdir -p $O
cd $ROOT
for u in 4 8 6 4; do
  touch pyimcom_amd/csrc/la_kernels.hip
  make -s -C pyimcom_amd/csrc EXTRA="-DIMCOM_EPI_U=$u" > $O/make_$u.log 2>&1 || { echo "build $u failed"; tail -5 $O/make_$u.log; exit 1; }
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-block --no-configs > $O/bench_u$u.json 2> $O/bench_u$u.err || { echo "bench $u failed"; tail -3 $O/bench_u$u.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$O/bench_u$u.json").read().strip().splitlines()[-1])
print("EPI_U=$u", round(d["value"], 1), "stamps/s", round(d["ms_per_step"], 2), "ms/step; epilogue", round(d["stage_ms_per_step"]["epilogue"], 3), "ms")
PY
done
touch pyimcom_amd/csrc/la_kernels.hip
