"""Block-sized table request (npsf^2 pairs from npsf spectra) with different intermediate chunk sizes (IMCOM_FFT_CHUNK_PAIRS):
    PYTHONPATH=. python tools/bench_table_chunks.py [npsf=110]"""
import os, sys
import numpy as np, torch
from pyimcom_amd._lib import default_context
from pyimcom_amd.stamps import overlap_tables, psf_spectra
npsf = int(sys.argv[1]) if len(sys.argv) > 1 else 110
ns, nfft = 383, 768
dev = torch.device("cuda:0")
ctx = default_context()
rng = np.random.default_rng(3)
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
for chunk in (None, 2048, 1024, 512, 256, 128, 96, 64, 32):
    if chunk is None: os.environ.pop("IMCOM_FFT_CHUNK_PAIRS", None)
    else: os.environ["IMCOM_FFT_CHUNK_PAIRS"] = str(chunk)
    ms = timed(lambda: overlap_tables(ctx, None, spec, None, spec, ns, nfft, pairs, None, out))
    print(f"chunk {chunk}: {ms:8.2f} ms for {len(pairs)} tables = {ms / len(pairs) * 1e3:.2f} us/table")
