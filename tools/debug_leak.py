import gc, sys, types, torch
sys.path.insert(0, ".")
import bench
from pyimcom_amd._lib import Context
from pyimcom_amd.blockrun import release_buffers
from pyimcom_amd.stamps import BlockTables, StampBatch
ctx = Context(0)
gc.disable()
r = bench.block_leg(ctx, "cuda:0", n1P=8, reps=1)
release_buffers(); ctx.release_workspace()
objs = [o for o in gc.get_objects() if isinstance(o, (BlockTables, StampBatch))]
print("alive:", [type(o).__name__ for o in objs])
def describe(o):
    if isinstance(o, types.FrameType): return f"frame {o.f_code.co_name}:{o.f_lineno}"
    if isinstance(o, types.FunctionType): return f"function {o.__qualname__}"
    if isinstance(o, types.CellType): return "cell"
    if isinstance(o, dict): return f"dict keys={list(o)[:6]}"
    if isinstance(o, (list, tuple)): return f"{type(o).__name__} len={len(o)}"
    return type(o).__name__
seen = set()
def walk(o, depth, path):
    if depth > 6 or id(o) in seen: return
    seen.add(id(o))
    for r_ in gc.get_referrers(o):
        if r_ is objs or isinstance(r_, types.FrameType) and r_.f_code.co_name in ("walk", "<module>", "<listcomp>"): continue
        print("  " * depth + describe(r_))
        walk(r_, depth + 1, path)
for o in objs[:1]:
    print("==", type(o).__name__)
    walk(o, 1, [])
