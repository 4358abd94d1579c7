"""Where the time of ONE stamp through the kernel-class seam (host buffers) goes: HIP-event families + wall clock."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from pyimcom_amd._lib import default_context  # noqa: E402
from pyimcom_amd.lakernel import HipCholKernel  # noqa: E402
from tests.golden.make_golden import make_outst  # noqa: E402

rng = np.random.default_rng(0)
n, m = 2208, 2304
pts = rng.uniform(0, 60, (n, 2)); outp = rng.uniform(5, 55, (m, 2))
A = np.exp(-((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1) / 3.0)
B = np.exp(-((outp[:, None, :] - pts[None, :, :]) ** 2).sum(-1) / 3.5)[None]
ctx = default_context()
for rep in range(4):
    o = make_outst(A, B, np.array([1.0]), 48, np.array([6e-4]), 1e-6, 0.5)
    if rep == 3:
        ctx.profile_enable(True)
        ctx.profile_reset()
    t = time.perf_counter(); HipCholKernel(o)(); dt = time.perf_counter() - t
print(f"wall {dt * 1e3:.2f} ms")
tot = 0.0
for fam in ("pack", "chol_gemm", "chol_diag", "solve_gemm", "solve_dinv", "finalize"):
    ms, nl = ctx.profile_get(fam)
    tot += ms
    print(f"  {fam:10s} {ms:7.3f} ms in {nl} launches")
print(f"  GPU kernels {tot:.2f} ms; the rest is PCIe (A {A.nbytes / 1e6:.0f} MB + B {B.nbytes / 1e6:.0f} MB in, T {B.size * 4 / 1e6:.0f} MB out) and host")
