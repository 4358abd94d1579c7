"""Host -> device and device -> host rates for 40 MB buffers: pageable numpy, page-locked torch, and pageable staged through page-locked
chunks by worker threads (what a seam that is handed pageable arrays can do):  python tools/bench_pcie.py"""
import sys, time, threading
import numpy as np, torch
sys.path.insert(0, ".")
n = 2208 * 2304
a = np.random.default_rng(0).standard_normal(n)
d = torch.empty(n, dtype=torch.float64, device="cuda")
pin = torch.empty(n, dtype=torch.float64, pin_memory=True)
def rate(f, reps=8):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return a.nbytes * reps / (time.perf_counter() - t) / 1e9
print("H2D pageable numpy  %.1f GB/s" % rate(lambda: d.copy_(torch.from_numpy(a), non_blocking=True)))
pin.numpy()[:] = a
print("H2D page-locked     %.1f GB/s" % rate(lambda: d.copy_(pin, non_blocking=True)))
print("host memcpy 1 thread %.1f GB/s" % rate(lambda: np.copyto(pin.numpy(), a)))
def staged(k=8, threads=4):
    chunks = np.array_split(np.arange(n), k)
    pv = pin.numpy()
    def work(q):
        for c in chunks[q::threads]:
            pv[c[0]:c[-1] + 1] = a[c[0]:c[-1] + 1]
    ts = [threading.Thread(target=work, args=(q,)) for q in range(threads)]
    [t.start() for t in ts]; [t.join() for t in ts]
    d.copy_(pin, non_blocking=True)
for th in (1, 2, 4, 8):
    print("H2D pageable -> pinned by %d threads -> device (not pipelined) %.1f GB/s" % (th, rate(lambda: staged(8, th))))
h = torch.empty(n, dtype=torch.float64)
print("D2H into pageable   %.1f GB/s" % rate(lambda: h.copy_(d)))
print("D2H into page-locked %.1f GB/s" % rate(lambda: pin.copy_(d, non_blocking=True)))
