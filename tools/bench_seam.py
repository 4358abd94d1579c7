import time, numpy as np, sys
sys.path.insert(0, '.')
from tests.golden.make_golden import make_outst
from pyimcom_amd.lakernel import HipCholKernel, HipEigenKernel
rng = np.random.default_rng(0)
n, m = 2208, 2304
pts = rng.uniform(0, 60, (n, 2)); outp = rng.uniform(5, 55, (m, 2))
A = np.exp(-((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1) / 3.0)
B = np.exp(-((outp[:, None, :] - pts[None, :, :]) ** 2).sum(-1) / 3.5)[None]
for K, kc in ((HipCholKernel, [6e-4]), (HipCholKernel, [1e-5, 1e-4, 1e-3])):
    for rep in range(3):
        o = make_outst(A, B, np.array([1.0]), 48, np.array(kc), 1e-6, 0.5)
        t = time.perf_counter(); K(o)(); dt = time.perf_counter() - t
    print(K.__name__, kc, f"{dt*1e3:.1f} ms per stamp through host buffers (PCIe-inclusive)")
n, m = 1024, 1024
A = A[:n, :n]; B = B[:, :m, :n]
for kc in ([6e-4], [1e-5, 1e-4, 1e-3]):
    for rep in range(2):
        o = make_outst(A, B, np.array([1.0]), 32, np.array(kc), 1e-6, 0.5)
        t = time.perf_counter(); HipEigenKernel(o)(); dt = time.perf_counter() - t
    print("HipEigenKernel N=1024 m=1024", kc, f"{dt*1e3:.1f} ms")
