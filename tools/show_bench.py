import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
print(json.dumps(d.get("block"), indent=0)[:1600])
print(json.dumps(d.get("cpu_baseline"))[:1400])
