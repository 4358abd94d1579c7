"""bench.py's paper4 leg alone (resident batches + the first passes of the production block):  python tools/bench_paper4.py [block_passes] [cpu_budget_s]"""
import json, sys
sys.path.insert(0, '.')
import bench
from pyimcom_amd._lib import Context

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cpu = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
ctx = Context(0)
out = bench.paper4_leg(ctx, "cuda:0", cpu_budget=cpu, block_passes=passes)
out.pop("telemetry", None)
print(json.dumps(out))
