"""Does the A builder inside a block run faster when the tables it reads were written a moment ago?  The bench's block workload
(a corner of n1P x n1P stamps, PSF group per 2 x 2 InStamps) with explicit passes of 16 / 64 / 256 stamps: stage times per stamp.
    PYTHONPATH=. python tools/ab_block_batch.py [n1P=32] [batches ...]"""
import sys, time
import torch
import bench
from pyimcom_amd import psfs as psfmod
from pyimcom_amd._lib import Context
from pyimcom_amd.blockrun import coadd_block
from pyimcom_amd.stamps import BlockTables

n1P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
batches = [int(a) for a in sys.argv[2:]] or [256, 64, 16, 256]
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
ctx = Context(0)
cfg, inst, pool, psfs, target, groups, counts, img_all, yxco_all = bench.block_workload(dev, n1P, config="cfg2")
E, ns = cfg.n_expo, psfs.shape[-1]
order = {k: q for q, k in enumerate(groups)}

def sample_groups(keys):
    idx = torch.tensor([order[k] for k in keys]).pin_memory().to(dev, non_blocking=True)
    im, yx = img_all[idx], yxco_all[idx]
    return psfmod.sample_psf(im.reshape(-1, ns + 16, ns + 16), ns, yx.reshape(-1, 2, ns, ns), psf_norm=True, ctx=ctx)

tabs = BlockTables(groups, target, cfg.nfft, ctx=ctx, device=dev, group_count=counts, bulk_provider=sample_groups, cells=True, eager_groups=True)
fams = ("psf_overlap", "build_A", "build_B", "chol_gemm", "solve_gemm", "epilogue")
coadd_block(cfg, pool, tabs, n1P, E, batch=256)
torch.cuda.synchronize()
ctx.profile_enable(True)
for b in batches:
    tabs.reset()
    ctx.profile_reset()
    t0 = time.perf_counter()
    coadd_block(cfg, pool, tabs, n1P, E, batch=b)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = {f: ctx.profile_get(f)[0] for f in fams}
    print(f"batch {b:4d}: {dt * 1e3:8.1f} ms per {n1P}x{n1P} block; us per stamp: " + "  ".join(f"{f} {v * 1e3 / (n1P * n1P):7.1f}" for f, v in st.items()), flush=True)
