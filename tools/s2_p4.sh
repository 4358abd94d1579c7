set -e
O=gpurun_out/s2; mkdir -p $O
for m in 1 0; do
IMCOM_LMIN_SKINNY=$m timeout -k 10 400 python tools/bench_paper4.py 4 > $O/p4_skinny$m.json 2> $O/p4_skinny$m.err || { tail -30 $O/p4_skinny$m.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/p4_skinny$m.json"))
def pick(x,keys):
    return {k:x[k] for k in keys if k in x}
print("skinny=$m", json.dumps({k:(v if not isinstance(v,dict) else {a:b for a,b in v.items() if not isinstance(b,(dict,list)) or a in ("stage_ms","stage_ms_per_step","pass_seconds")}) for k,v in d.items() if k in ("value","ms_per_stamp","stage_ms_per_step","block","roofline")})[:1800])
PY
done
