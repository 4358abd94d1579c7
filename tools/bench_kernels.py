"""Resident path per LA kernel at cfg-2 geometry:  PYTHONPATH=. python tools/bench_kernels.py [kernel=Iterative] [batch=16] [inpad_as]
prints ms per stamp of build / solve / coadd (wall clock around synchronised phases)."""
import dataclasses, sys, time
import numpy as np
import torch
from pyimcom_amd import synth
from pyimcom_amd._lib import default_context
from pyimcom_amd.stamps import PSFGroupTables, StampBatch

kernel = sys.argv[1] if len(sys.argv) > 1 else "Iterative"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
kw = dict(kernel=kernel)
if kernel in ("Iterative", "Empirical"):
    kw["kappaC"] = (0.0,) if kernel == "Iterative" else (6e-4,)
if len(sys.argv) > 3:
    kw["inpad_as"] = float(sys.argv[3])
cfg = dataclasses.replace(synth.CONFIGS["cfg2"], **kw)
ctx = default_context()
stamps = [synth.make_stamp(cfg, i) for i in range(batch)]
psfs, target = synth.make_psfs(cfg, cfg.n_expo)
tables = PSFGroupTables(psfs, target, cfg.nfft, ctx=ctx)
sb = StampBatch(cfg, stamps, tables, ctx=ctx)
sb.run(); torch.cuda.synchronize()
t = {}
for name, fn in (("build", sb.build), ("solve", sb.solve), ("coadd", sb.coadd)):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); t[name] = (time.perf_counter() - t0) / batch * 1e3
print(f"{kernel} batch {batch} N~{int(np.mean(sb.n))}: " + ", ".join(f"{k} {v:.2f} ms/stamp" for k, v in t.items()))
