"""cProfile of the device loop's thread in refblock.coadd_output_stamps (duck-typed 16 x 16-stamp block, host halves on worker threads):
    python tools/profile_refblock.py [threads=16]"""
import cProfile, pstats, sys, time
sys.path.insert(0, ".")
import torch
from pyimcom_amd import synth
from pyimcom_amd._lib import default_context
from pyimcom_amd.refblock import coadd_output_stamps
thr = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = synth.CONFIGS["cfg2"]
blk, psfgrp, _, _ = synth.duck_block(cfg, 16, cfg.n_expo, seed=5)
ctx = default_context()
coadd_output_stamps(blk, psfgrp, ctx=ctx, host_threads=thr)
torch.cuda.synchronize()
pr = cProfile.Profile()
t = time.perf_counter()
pr.enable()
coadd_output_stamps(blk, psfgrp, ctx=ctx, host_threads=thr)
torch.cuda.synchronize()
pr.disable()
print("wall %.1f ms" % ((time.perf_counter() - t) * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
