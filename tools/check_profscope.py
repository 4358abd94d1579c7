"""ProfScope family time vs torch events vs wall clock for one large table request (debug aid)."""
import time
import numpy as np
import torch
from pyimcom_amd import synth
from pyimcom_amd._lib import default_context
from pyimcom_amd.stamps import BlockTables

cfg = synth.CONFIGS["cfg2"]
E = cfg.n_expo
psfs, target = synth.make_psfs(cfg, E)
ng = 5
groups = {(gj, gi): psfs for gj in range(ng) for gi in range(ng)}
ctx = default_context()
for rep in range(3):
    bt = BlockTables(groups, target, cfg.nfft, capacity=6000, ctx=ctx)
    keys = []
    for gj in range(ng - 1):
        for gi in range(ng - 1):
            keys += BlockTables.keys_for([(gj, gi), (gj, gi + 1), (gj + 1, gi), (gj + 1, gi + 1)])
    torch.cuda.synchronize()
    ctx.profile_enable(True)
    ctx.profile_reset()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    bt.require(keys)
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"tables {bt.used}: ProfScope psf_overlap {ctx.profile_get('psf_overlap')[0]:.2f} ms, spectra {ctx.profile_get('psf_spectra')[0]:.2f} ms, "
          f"torch events {e0.elapsed_time(e1):.2f} ms, host enqueue {(t1 - t0) * 1e3:.2f} ms, wall {(t2 - t0) * 1e3:.2f} ms")
    ctx.profile_enable(False)
