"""Timing of the eigen path on the device-resident stamp pipeline (cfg-3 style)."""
import sys, time
sys.path.insert(0, '.')
import dataclasses
import numpy as np, torch
from pyimcom_amd import synth
from pyimcom_amd.stamps import PSFGroupTables, StampBatch
name, nb = sys.argv[1], int(sys.argv[2])
cfg = synth.CONFIGS[name]
if len(sys.argv) > 3:
    cfg = dataclasses.replace(cfg, n2=int(sys.argv[3]))
stamps = [synth.make_stamp(cfg, i) for i in range(nb)]
psfs, target = synth.make_psfs(cfg, max(s.n_expo for s in stamps))
tabs = PSFGroupTables(psfs, target, cfg.nfft)
b = StampBatch(cfg, stamps, tabs)
b.ctx.profile_enable(True)
for rep in range(2):
    b.ctx.profile_reset()
    torch.cuda.synchronize(); t = time.perf_counter()
    b.run(); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"{name} batch {nb} N~{b.n.mean():.0f}: {dt*1e3:.1f} ms/step = {dt/nb*1e3:.1f} ms/stamp", {f: round(b.ctx.profile_get(f)[0], 1) for f in ("eigen_jacobi", "eigen_trd", "eigen_orgtr", "eigen_qr", "eigen_gemm", "lakernel1", "build_A", "build_B")})
