"""Time the PSF-overlap table construction (imcom_psf_overlap) at cfg-2 geometry: the self + input-output set of one
PSF group (E(E+1)/2 + E tables) and a cross set between two groups (E^2 tables).  python bench_tables.py [E] [reps]"""
import sys
import time

import numpy as np
import torch

from pyimcom_amd import synth
from pyimcom_amd.stamps import BlockTables, PSFGroupTables

E = int(sys.argv[1]) if len(sys.argv) > 1 else 6
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = synth.CONFIGS["cfg2"]
psfs, target = synth.make_psfs(cfg, E)
p = torch.as_tensor(psfs, device="cuda:0")
t = torch.as_tensor(target, device="cuda:0")
PSFGroupTables(p, t, cfg.nfft)  # warm-up (workspace allocation)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    tabs = PSFGroupTables(p, t, cfg.nfft)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
nt = tabs.tables.shape[0] + 1
print(f"group set: {nt} tables in {dt * 1e3:.2f} ms = {dt / nt * 1e6:.1f} us/table")
bt = BlockTables({(0, 0): psfs, (0, 1): psfs[::-1].copy()}, target, cfg.nfft, capacity=4 * E * E + 64)
bt.require([("cross", (0, 0), (0, 1))])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    bt.drop_all()
    bt.require([("cross", (0, 0), (0, 1))])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"cross set: {E * E} tables in {dt * 1e3:.2f} ms = {dt / (E * E) * 1e6:.1f} us/table")
