import sys, time, threading
sys.path.insert(0, '.')
import numpy as np, torch
from pyimcom_amd import synth
from pyimcom_amd._lib import Context
from pyimcom_amd.stamps import PSFGroupTables, StampBatch
cfg = synth.CONFIGS[sys.argv[3] if len(sys.argv) > 3 else "cfg2"]
nthr, per = int(sys.argv[1]), int(sys.argv[2])
psfs, target = synth.make_psfs(cfg, cfg.n_expo)
ctxs = [Context(0) for _ in range(nthr)]
streams = [torch.cuda.Stream() for _ in range(nthr)]
batches = []
for t in range(nthr):
    with torch.cuda.stream(streams[t]):
        tabs = PSFGroupTables(psfs, target, cfg.nfft, ctx=ctxs[t])
        batches.append(StampBatch(cfg, [synth.make_stamp(cfg, t * per + i) for i in range(per)], tabs, ctx=ctxs[t]))
torch.cuda.synchronize()
def work(t, steps):
    with torch.cuda.stream(streams[t]):
        for _ in range(steps):
            batches[t].run()
        streams[t].synchronize()
for steps in (1, 3):
    ths = [threading.Thread(target=work, args=(t, steps)) for t in range(nthr)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    [th.start() for th in ths]; [th.join() for th in ths]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"threads {nthr} x batch {per}: {nthr*per*steps/dt:.1f} stamps/s ({dt/steps*1e3:.1f} ms/step)")
