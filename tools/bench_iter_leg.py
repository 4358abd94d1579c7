"""bench.py's iter_default leg alone:  python tools/bench_iter_leg.py [cpu_budget_s] [batch]"""
import json, sys
sys.path.insert(0, '.')
import bench
from pyimcom_amd._lib import Context

cpu = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ctx = Context(0)
print(json.dumps(bench.iter_default_leg(ctx, "cuda:0", batch=batch, cpu_budget=cpu)))
