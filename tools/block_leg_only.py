"""bench.py's block leg alone (for traces):  python tools/block_leg_only.py [reps=2]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from pyimcom_amd._lib import Context
torch.cuda.set_device(0)
ctx = Context(0)
print(json.dumps(bench.block_leg(ctx, torch.device("cuda:0"), reps=int(sys.argv[1]) if len(sys.argv) > 1 else 2)))
