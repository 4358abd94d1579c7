"""Where the host half of a PSF group goes (refblock.input_psf_groups: PSF broker + WCS evaluation of the sampling positions, then the
adapter's own array work), on the bench's duck-typed block:  python tools/profile_host_half.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from pyimcom_amd import synth
cfg = synth.CONFIGS["cfg2"]
blk, psfgrp, _, _ = synth.duck_block(cfg, 16, cfg.n_expo, seed=5)
ns = int(psfgrp.nsamp)
lin = np.arange(ns) - (ns - 1) / 2.0
gx, gy = np.meshgrid(lin, lin)
xy = np.stack([gx.ravel(), gy.ravel()], axis=1) * float(psfgrp.dscale)
p0 = np.array(blk.instamps[2][2].psf_compute_point_pix, dtype=np.float64)
world = blk.outwcs.all_pix2world(np.array([p0]), 0)[0]
im = blk.inimages[0]
def t(f, n=20):
    f(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    return (time.perf_counter() - t0) / n * 1e3, r
ms, img = t(lambda: np.asarray(im.get_psf_pos(world, use_shortrange=True), dtype=np.float64)); print(f"get_psf_pos            {ms:6.2f} ms")
ms, a = t(lambda: im.outpix2world2inpix(xy + p0)); print(f"outpix2world2inpix(xy) {ms:6.2f} ms  (callee, incl. xy + p0)")
ms, _ = t(lambda: xy + p0); print(f"  of which xy + p0     {ms:6.2f} ms")
b = np.asarray(im.outpix2world2inpix(p0[None]))
ms, d = t(lambda: (np.asarray(a) - b) * 8.0); print(f"(a - b) * oversamp     {ms:6.2f} ms")
yx = torch.empty((6, 2, ns, ns), dtype=torch.float64, pin_memory=True); yv = yx.numpy()
def fill():
    yv[0, 0], yv[0, 1] = d[:, 1].reshape(ns, ns), d[:, 0].reshape(ns, ns)
ms, _ = t(fill); print(f"into the pinned stack  {ms:6.2f} ms")
imgs = torch.empty((6,) + img.shape, dtype=torch.float64, pin_memory=True); iv = imgs.numpy()
def fill2(): iv[0] = img
ms, _ = t(fill2); print(f"image into pinned      {ms:6.2f} ms")
ms, _ = t(lambda: torch.empty((6, 2, ns, ns), dtype=torch.float64, pin_memory=True), 5); print(f"pinned alloc [6,2,ns,ns] {ms:6.2f} ms")
