"""End-to-end block throughput of the device-resident chain (InStamp pool -> selection -> tables -> A, B -> Cholesky
-> coaddition -> block maps) at cfg-2 geometry: a block of n1P x n1P output stamps, (a) one PSF group for the block,
(b) the reference's PSF group per 2x2 InStamps (BlockTables: self / input-output / cross table sets on demand).
    python bench_block.py [n1P=16] [batch=64]"""
import sys
import time

import numpy as np
import torch

from pyimcom_amd import synth
from pyimcom_amd.blockrun import coadd_block
from pyimcom_amd.select import InStampPool
from pyimcom_amd.stamps import BlockTables, PSFGroupTables

n1P = int(sys.argv[1]) if len(sys.argv) > 1 else 16
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = synth.CONFIGS["cfg2"]
E = cfg.n_expo
rng = np.random.default_rng(5)
inst = synth.make_instamps(cfg, n1P, E, rng)
pool = InStampPool(inst, cfg.n_inframe)
psfs, target = synth.make_psfs(cfg, E)
nst = n1P + 2
print(f"block {n1P}x{n1P} stamps, {pool.npool} input pixels in {nst * nst} InStamps, batch {batch}")


def timed(label, tables_factory, reps=3, pipeline=True):
    tabs = tables_factory()
    coadd_block(cfg, pool, tabs, n1P, E, batch=batch, pipeline=pipeline)  # warm-up: workspace and buffers
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        tabs = tables_factory()  # table construction is part of the block
        maps = coadd_block(cfg, pool, tabs, n1P, E, batch=batch, pipeline=pipeline)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{label}: {dt * 1e3:8.1f} ms per block = {n1P * n1P / dt:7.1f} stamps/s  (out_map rms {float(maps.out_map.square().mean().sqrt()):.4g})")


timed("one PSF group      ", lambda: PSFGroupTables(psfs, target, cfg.nfft))
ng = (nst + 1) // 2
lin = np.arange(psfs.shape[-1]) - psfs.shape[-1] // 2
groups = {}
for gj in range(ng):
    for gi in range(ng):
        mod = 1.0 + 0.02 * np.sin(0.05 * lin * (1 + gi % 3))[None, None, :] + 0.02 * np.cos(0.04 * lin * (1 + gj % 3))[None, :, None]
        q = psfs * mod
        groups[(gj, gi)] = q / q.sum(axis=(1, 2), keepdims=True)
for pl in (True, False, True, False):
    timed(f"{ng * ng} PSF groups (2x2), pipeline={pl}", lambda: BlockTables(groups, target, cfg.nfft, capacity=13500), pipeline=pl)
