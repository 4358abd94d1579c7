"""Overlap-table path alone (PSF spectra, then tables from spectra pairs) at a PSFGrp geometry:
    PYTHONPATH=. python tools/bench_fft_lines.py [npixpsf=48] [oversamp=8] [npsf=12] [reps=5]
prints us per spectrum and us per table (npsf^2 tables per call) with HIP events, and checks one table against numpy."""
import sys
import time

import numpy as np
import torch

from pyimcom_amd._lib import default_context
from pyimcom_amd.stamps import overlap_tables, psf_spectra

npixpsf = int(sys.argv[1]) if len(sys.argv) > 1 else 48
oversamp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
npsf = int(sys.argv[3]) if len(sys.argv) > 3 else 12
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
ns, nfft = npixpsf * oversamp - 1, npixpsf * oversamp * 2
dev = torch.device("cuda:0")
ctx = default_context()
rng = np.random.default_rng(3)
yy, xx = np.mgrid[:ns, :ns] - ns // 2
psf = np.stack([np.exp(-(xx**2 + yy**2) / (2.0 * (3.0 + 0.2 * k) ** 2)) + 0.01 * rng.standard_normal((ns, ns)) for k in range(npsf)])
p = torch.as_tensor(psf, device=dev)
pairs = np.array([(i, j) for i in range(npsf) for j in range(npsf)], dtype=np.int32)
out = torch.empty((len(pairs), ns + 12, ns + 12), dtype=torch.float64, device=dev)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


spec = psf_spectra(ctx, p, nfft)
t_spec = timed(lambda: psf_spectra(ctx, p, nfft))
t_tab = timed(lambda: overlap_tables(ctx, p, spec, p, spec, ns, nfft, pairs, None, out))
print(f"nsamp {ns} nfft {nfft}: {t_spec / npsf:.2f} us per spectrum, {t_tab / len(pairs):.2f} us per table ({len(pairs)} tables per call)")
# numpy check of table (1, 2)
t0 = time.perf_counter()
f = np.zeros((2, nfft, nfft))
f[:, :ns, :ns] = psf[1:3]
r = np.fft.rfft2(f)
full = np.fft.irfft2(r[0] * np.conj(r[1]), s=(nfft, nfft))
ref = np.roll(full, (ns // 2, ns // 2), axis=(0, 1))[:ns, :ns]
got = out[1 * npsf + 2].cpu().numpy()
err = np.abs(got[6:-6, 6:-6] - ref).max() / np.abs(ref).max()
border = max(np.abs(got[:6]).max(), np.abs(got[-6:]).max(), np.abs(got[:, :6]).max(), np.abs(got[:, -6:]).max())
print(f"table (1,2) vs numpy: rel err {err:.2e}, border max {border:.1e}")
assert err < 2e-13 and border == 0.0
