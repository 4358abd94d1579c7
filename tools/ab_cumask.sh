#!/bin/bash
# A/B on one box: the Eigen path (cfg-3) with the second queue / the sub-batches' streams confined to a share of the CUs.
#   tools/ab_cumask.sh   (under gpurun; results in gpurun_out/r04_cumask/)
set -e
out=gpurun_out/r04_cumask; mkdir -p $out
run() { # name batch env...
    local name=$1 b=$2; shift 2
    env "$@" python tools/bench_eigen.py cfg3 $b > $out/$name.json 2> $out/$name.err || { tail -5 $out/$name.err; return 1; }
    python - "$out/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-28s %7.3f ms/stamp  trd %7.1f applyq %6.1f la %5.1f" % (sys.argv[2], d["ms_per_stamp"], d["stage_ms_per_step"]["eigen_trd"], d["stage_ms_per_step"]["eigen_applyq"], d["stage_ms_per_step"]["lakernel1"]), flush=True)
PY
}
run b32_default 32 X=1
run b32_prio 32 IMCOM_SPLIT_PRIO=1
run b64_default 64 X=1
run b64_prio 64 IMCOM_SPLIT_PRIO=1
run b256_default 256 X=1
run b256_prio 256 IMCOM_SPLIT_PRIO=1
run b256_split3_prio 256 IMCOM_SPLIT_PRIO=1 IMCOM_EIGEN_SPLIT=3
run b32_default_b 32 X=1
