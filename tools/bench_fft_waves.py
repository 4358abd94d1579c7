"""Block-sized table request (npsf^2 pairs from npsf spectra) with 4 / 8 / 12 waves per workgroup of the line FFTs (IMCOM_FFT_WAVES):
how the per-table time scales with the waves in flight tells whether the transforms wait for latency or for a pipe.
    PYTHONPATH=. python tools/bench_fft_waves.py [npsf=60]"""
import os, sys
import numpy as np, torch
from pyimcom_amd._lib import default_context
from pyimcom_amd.stamps import overlap_tables, psf_spectra
npsf = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ns, nfft = 383, 768
dev = torch.device("cuda:0")
ctx = default_context()
yy, xx = np.mgrid[:ns, :ns] - ns // 2
base = np.exp(-(xx**2 + yy**2) / (2.0 * 3.0**2))
p = torch.as_tensor(np.stack([base * (1 + 0.01 * k) for k in range(npsf)]), device=dev)
pairs = np.array([(i, j) for i in range(npsf) for j in range(npsf)], dtype=np.int32)
out = torch.empty((len(pairs), ns + 12, ns + 12), dtype=torch.float64, device=dev)
spec = psf_spectra(ctx, p, nfft)
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for w in (sys.argv[2:] or ["12", "8", "4", "12"]):
    os.environ["IMCOM_FFT_WAVES"] = w
    ms = timed(lambda: overlap_tables(ctx, None, spec, None, spec, ns, nfft, pairs, None, out))
    print(f"waves {w}: {ms:8.2f} ms for {len(pairs)} tables = {ms / len(pairs) * 1e3:.3f} us/table", flush=True)
