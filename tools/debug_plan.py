import numpy as np, torch
from pyimcom_amd import synth
from pyimcom_amd.blockrun import available_bytes, coadd_block, pass_bytes, plan_block, release_buffers, _BUFS, _SIDE
from pyimcom_amd.select import InStampPool
from pyimcom_amd.stamps import NB, PSFGroupTables, free_device_bytes
cfg = synth.CONFIGS["cfg2"]; n1P, E = 8, cfg.n_expo
inst = synth.make_instamps(cfg, n1P, E, np.random.default_rng(8))
pool = InStampPool(inst, cfg.n_inframe)
psfs, target = synth.make_psfs(cfg, E)
tabs = PSFGroupTables(psfs, target, cfg.nfft); ctx = tabs.ctx
def show(tag):
    torch.cuda.synchronize()
    f, t = torch.cuda.mem_get_info()
    ws = 0 if ctx._ws is None else ctx._ws.numel()
    bufs = sum(b.nbytes() for v in _BUFS.values() for b in v)
    side = sum((0 if c._ws is None else c._ws.numel()) for _, c in _SIDE.values())
    print(f"{tag}: free {f/2**20:.1f} MiB reserved {torch.cuda.memory_reserved()/2**20:.1f} allocated {torch.cuda.memory_allocated()/2**20:.1f} ws {ws/2**20:.1f} bufs {bufs/2**20:.1f} side_ws {side/2**20:.1f} avail {available_bytes(pool.device, ctx)/2**20:.1f}")
show("start")
ldm = (cfg.m + NB - 1) // NB * NB
room = pass_bytes(24, 2304, ldm, 1, "Cholesky", nv=1, n_inframe=cfg.n_inframe) + (3 << 30)
spare = available_bytes(pool.device, ctx) - room
hog = torch.empty(spare, dtype=torch.uint8, device=pool.device)
show("hog")
print([len(c) for c in plan_block(cfg, pool, tabs, n1P)])
show("planned")
maps = coadd_block(cfg, pool, tabs, n1P, E)
show("ran")
print([len(c) for c in plan_block(cfg, pool, tabs, n1P)])
del maps
show("del maps")
print(torch.cuda.memory_summary()[:3000])
