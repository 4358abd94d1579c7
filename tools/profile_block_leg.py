"""cProfile of the host side of bench.py's block leg (one 48 x 48 block):  python tools/profile_block_leg.py [n1P]"""
import cProfile, json, os, pstats, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from pyimcom_amd._lib import Context
torch.cuda.set_device(0)
ctx = Context(0)
n1P = int(sys.argv[1]) if len(sys.argv) > 1 else 48
pr = cProfile.Profile()
pr.enable()
r = bench.block_leg(ctx, torch.device("cuda:0"), n1P=n1P, reps=1)
pr.disable()
print(json.dumps(r))
pstats.Stats(pr).sort_stats("cumulative").print_stats(60)
