"""The A builder inside a block: is it the NUMBER of tables a stamp's samples are spread over?  The bench's block workload with the same PSF
images in every 2 x 2 group, coadded (a) with a PSF group per 2 x 2 InStamps (every stamp reads ~300 distinct table copies of four groups)
and (b) with ONE group's 27 tables for the whole block -- same pixels, same separations, same table contents.
    PYTHONPATH=. python tools/ab_block_tables.py [n1P=32]"""
import sys, time
import torch
import bench
from pyimcom_amd import psfs as psfmod
from pyimcom_amd._lib import Context
from pyimcom_amd.blockrun import coadd_block
from pyimcom_amd.stamps import BlockTables, PSFGroupTables

n1P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
ctx = Context(0)
cfg, inst, pool, psfs, target, groups, counts, img_all, yxco_all = bench.block_workload(dev, n1P, identical=True, config="cfg2")
E, ns = cfg.n_expo, psfs.shape[-1]
order = {k: q for q, k in enumerate(groups)}

def sample_groups(keys):
    idx = torch.tensor([order[k] for k in keys]).pin_memory().to(dev, non_blocking=True)
    return psfmod.sample_psf(img_all[idx].reshape(-1, ns + 16, ns + 16), ns, yxco_all[idx].reshape(-1, 2, ns, ns), psf_norm=True, ctx=ctx)

fams = ("psf_overlap", "build_A", "build_B", "chol_gemm", "solve_gemm", "epilogue")
grp = BlockTables(groups, target, cfg.nfft, ctx=ctx, device=dev, group_count=counts, bulk_provider=sample_groups, cells=True, eager_groups=True)
one = PSFGroupTables(sample_groups([(0, 0)]), target, cfg.nfft, ctx=ctx, device=dev)
for name, tabs in (("group per 2x2 InStamps", grp), ("one group", one), ("group per 2x2 InStamps", grp), ("one group", one)):
    if tabs is grp:
        tabs.reset()
    coadd_block(cfg, pool, tabs, n1P, E, batch=256) if name is None else None
    torch.cuda.synchronize()
    ctx.profile_enable(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    coadd_block(cfg, pool, tabs, n1P, E, batch=256)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = {f: ctx.profile_get(f)[0] for f in fams}
    ctx.profile_enable(False)
    print(f"{name:24s}: {dt * 1e3:8.1f} ms per {n1P}x{n1P} block; us per stamp: " + "  ".join(f"{f} {v * 1e3 / (n1P * n1P):7.1f}" for f, v in st.items()), flush=True)
