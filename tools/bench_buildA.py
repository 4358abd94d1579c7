"""Time imcom_build_A alone on the cfg-2 batch (developer tool): python tools/bench_buildA.py [batch] [reps]."""
import sys, time
sys.path.insert(0, '.')
import torch
from pyimcom_amd import synth
from pyimcom_amd.stamps import PSFGroupTables, StampBatch
cfg = synth.CONFIGS[sys.argv[3] if len(sys.argv) > 3 else "cfg2"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
stamps = [synth.make_stamp(cfg, i) for i in range(B)]
psfs, target = synth.make_psfs(cfg, max(s.n_expo for s in stamps))
sb = StampBatch(cfg, stamps, PSFGroupTables(psfs, target, cfg.nfft))
import ctypes as C
from pyimcom_amd._lib import lib, check
from pyimcom_amd.stamps import _dp, _hp
def buildA():
    sb._stream()
    check(lib.imcom_build_A(sb.ctx.handle, sb.batch, _hp(sb.n), sb.ldn, _dp(sb.x), _dp(sb.y), _dp(sb.psf), _dp(sb.tables.tables),
                            sb.tables.tables.shape[0], C.byref(sb.geom), _dp(sb.pair_tab), _dp(sb.pair_pen), sb.npsf, _dp(sb.A)))
buildA(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    buildA()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"build_A: {dt*1e3:.2f} ms per {B} stamps = {dt/B*1e6:.1f} us/stamp")
