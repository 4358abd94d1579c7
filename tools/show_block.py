"""Headline and block leg of a bench.py JSON line:  python tools/show_block.py FILE"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", round(d["value"], 1), "stamps/s", round(d["ms_per_step"], 2), "ms/step frac", round(d["roofline"]["frac"], 4))
b = d.get("block")
if b:
    print("block", round(b["value"], 1), "stamps/s", round(b["ms_per_block"], 2), "ms batch", b["batch"], {k: round(v, 2) for k, v in b["stage_ms_per_block"].items()},
          "host+gaps", round(b["host_and_gaps_ms_per_block"], 2), "rms", b["out_map_rms"])
