"""cProfile of the host side of bench.py's block leg (warm-up + 2 timed blocks):  python tools/profile_block_leg.py"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from pyimcom_amd._lib import Context
torch.cuda.set_device(0)
ctx = Context(0)
bench.block_leg(ctx, torch.device("cuda:0"), reps=1)
pr = cProfile.Profile()
pr.enable()
r = bench.block_leg(ctx, torch.device("cuda:0"), reps=2)
pr.disable()
print(r["ms_per_block"], r["host_and_gaps_ms_per_block"])
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
