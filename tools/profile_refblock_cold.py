"""The FIRST block of a process through refblock.coadd_output_stamps (the reference runs one block per process): cProfile of the device
loop's thread and wall time.   python tools/profile_refblock_cold.py [n1P=48] [threads=16]"""
import cProfile, pstats, sys, time
sys.path.insert(0, ".")
t_imp = time.perf_counter()
import torch
from pyimcom_amd import synth
from pyimcom_amd._lib import default_context
from pyimcom_amd.refblock import coadd_output_stamps
n1P = int(sys.argv[1]) if len(sys.argv) > 1 else 48
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cfg = synth.CONFIGS["cfg2"]
blk, psfgrp, _, _ = synth.duck_block(cfg, n1P, cfg.n_expo, seed=5)
print(f"imports + duck block {time.perf_counter() - t_imp:.2f} s", flush=True)
t = time.perf_counter()
ctx = default_context()
torch.cuda.synchronize()
print(f"context {time.perf_counter() - t:.2f} s", flush=True)
pr = cProfile.Profile()
t = time.perf_counter()
pr.enable()
coadd_output_stamps(blk, psfgrp, ctx=ctx, host_threads=thr)
torch.cuda.synchronize()
pr.disable()
print(f"first block: wall {(time.perf_counter() - t) * 1e3:.0f} ms", flush=True)
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
