"""Summary of a bench.py JSON line:  python tools/show_bench.py FILE"""
import json, sys
d = json.load(open(sys.argv[1]))
print("value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2), "frac", round(d["roofline"]["frac"], 4), "probe", round(d["roofline"].get("mfma_probe_tflops", 0), 1))
print("stages", {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()})
for k in ("kernel_seam", "block_seam"):
    if k in d:
        print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in d[k].items() if a != "what"})
if "block" in d:
    b = d["block"]
    print("block", round(b["value"], 1), "ms", round(b["ms_per_block"], 1), "host+gaps", round(b["host_and_gaps_ms_per_block"], 1), b.get("batches"), b.get("tables"))
    print("block stages", {k: round(v, 1) for k, v in b["stage_ms_per_block"].items()})
if "cpu_baseline" in d:
    print("cpu", round(d["cpu_baseline"]["value"], 2), d["cpu_baseline"]["sample"])
