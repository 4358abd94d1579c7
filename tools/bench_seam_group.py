import sys, time, numpy as np
sys.path.insert(0, '.')
from tests.golden.make_golden import make_outst
from pyimcom_amd.lakernel import HipCholKernel, solve_chol_stamps
rng = np.random.default_rng(0)
n, m = 2208, 2304
pts = rng.uniform(0, 60, (n, 2)); outp = rng.uniform(5, 55, (m, 2))
A = np.exp(-((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1) / 3.0)
B = np.exp(-((outp[:, None, :] - pts[None, :, :]) ** 2).sum(-1) / 3.5)[None]
for kc in ([6e-4], [1e-5, 1e-4, 1e-3]):
    for rep in range(3):
        o = make_outst(A, B, np.array([1.0]), 48, np.array(kc), 1e-6, 0.5)
        t = time.perf_counter(); HipCholKernel(o)(); dt = time.perf_counter() - t
    for g in (2, 4, 8):
        for rep in range(3):
            os_ = [make_outst(A, B, np.array([1.0]), 48, np.array(kc), 1e-6, 0.5) for _ in range(g)]
            t = time.perf_counter(); solve_chol_stamps(os_); dg = time.perf_counter() - t
        print(kc, f"single {dt*1e3:.2f} ms; group of {g}: {dg*1e3/g:.2f} ms per stamp", flush=True)
