O=gpurun_out/s2; mkdir -p $O
for fm in 100000 8; do
IMCOM_LMIN_FEW_MAX=$fm timeout -k 10 400 python tools/bench_paper4.py 4 > $O/p4_few$fm.json 2> $O/p4_few$fm.err || { tail -30 $O/p4_few$fm.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/p4_few$fm.json"))
b=d["block"]; print("few_max=$fm batch", round(d["value"],1), d["stage_ms_per_step"]["eigen_repair"], "block", b["pass_seconds"], round(b["seconds_per_block"],2), round(b["value"],1), b["stage_ms"]["eigen_repair"])
PY
done
