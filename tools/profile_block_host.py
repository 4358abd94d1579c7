"""cProfile of the host side of one block with PSF groups (tools/bench_block.py geometry)."""
import cProfile
import pstats
import sys

import numpy as np
import torch

from pyimcom_amd import synth
from pyimcom_amd.blockrun import coadd_block
from pyimcom_amd.select import InStampPool
from pyimcom_amd.stamps import BlockTables

n1P = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = synth.CONFIGS["cfg2"]
E = cfg.n_expo
inst = synth.make_instamps(cfg, n1P, E, np.random.default_rng(5))
pool = InStampPool(inst, cfg.n_inframe)
psfs, target = synth.make_psfs(cfg, E)
ng = (n1P + 3) // 2
groups = {(gj, gi): psfs for gj in range(ng) for gi in range(ng)}
coadd_block(cfg, pool, BlockTables(groups, target, cfg.nfft, capacity=4096), n1P, E, batch=64)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
coadd_block(cfg, pool, BlockTables(groups, target, cfg.nfft, capacity=4096), n1P, E, batch=64)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
