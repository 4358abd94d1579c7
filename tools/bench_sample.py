"""PSF sampling alone (imcom_sample_psf, 6 PSFs of a cfg-2 group, rotated positions): PYTHONPATH=. python tools/bench_sample.py"""
import numpy as np
import torch
from pyimcom_amd import psfs as psfmod, synth
from pyimcom_amd._lib import default_context

cfg = synth.CONFIGS["cfg2"]
E = cfg.n_expo
p, _ = synth.make_psfs(cfg, E)
ns = p.shape[-1]
dev = torch.device("cuda:0")
ctx = default_context()
img = torch.zeros((E, ns + 16, ns + 16), dtype=torch.float64, device=dev)
img[:, 8 : 8 + ns, 8 : 8 + ns] = torch.as_tensor(p, device=dev)
lt = torch.arange(ns, dtype=torch.float64, device=dev) - (ns - 1) / 2.0
yo, xo = torch.meshgrid(lt, lt, indexing="ij")
th = torch.as_tensor([0.004 * (e - E / 2) for e in range(E)], dtype=torch.float64, device=dev)
c, sn = torch.cos(th)[:, None, None], torch.sin(th)[:, None, None]
yxco = torch.stack([c * yo + sn * xo, -sn * yo + c * xo], dim=1).contiguous()
for _ in range(3):
    out = psfmod.sample_psf(img, ns, yxco, psf_norm=True, ctx=ctx)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    out = psfmod.sample_psf(img, ns, yxco, psf_norm=True, ctx=ctx)
e1.record()
torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) / 20 * 1e3:.1f} us per group of {E} PSFs; sums {out.sum(dim=(1, 2)).cpu().numpy()}")
