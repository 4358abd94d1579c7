set -e
O=gpurun_out/s2; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_paper4.py tests/test_gpu_kernels_block.py tests/test_gpu_blockrun.py tests/test_gpu_farm.py -x -q -k "repair or paper4 or hint" > $O/tests_rep.log 2>&1 || { tail -30 $O/tests_rep.log; exit 1; }
tail -2 $O/tests_rep.log
IMCOM_LMIN_DEBUG=1 timeout -k 10 400 python tools/bench_paper4.py 4 > $O/p4_sk.json 2> $O/p4_sk.err || { tail -30 $O/p4_sk.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/p4_sk.json"))
print("batch leg", d["value"], d["stage_ms_per_step"]["eigen_repair"])
b=d["block"]; print("block", b["pass_seconds"], b["seconds_per_block"], b["value"], b["stage_ms"]["eigen_repair"])
PY
grep -E "^\[lmin\] [0-9]+ stamps|stamps ran" $O/p4_sk.err | tail -12
