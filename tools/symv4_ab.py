"""symv4 alone (profile level 2: HIP events around every launch), cfg-3, one stream:  python tools/symv4_ab.py 256"""
import json, os, sys
sys.path.insert(0, '.')
os.environ["IMCOM_EIGEN_SPLIT"] = "1"
import numpy as np, torch
from pyimcom_amd import synth
from pyimcom_amd.stamps import PSFGroupTables, StampBatch
nb = int(sys.argv[1])
cfg = synth.CONFIGS["cfg3"]
stamps = [synth.make_stamp(cfg, i) for i in range(nb)]
psfs, target = synth.make_psfs(cfg, 8)
b = StampBatch(cfg, stamps, PSFGroupTables(psfs, target, cfg.nfft))
b.build(); b.solve(); torch.cuda.synchronize()
b.ctx.profile_enable(2); b.ctx.profile_reset()
b.solve(); torch.cuda.synchronize()
ms, l = b.ctx.profile_get("symv4"); trd = b.ctx.profile_get("eigen_trd")[0]
n = b.n.astype(np.float64)
print(json.dumps({"batch": nb, "symv4_ms": round(ms, 1), "launches": l, "TBs": round(float((n**3 / 3).sum()) / (ms * 1e-3) / 1e12, 3), "trd_ms": round(trd, 1)}))
