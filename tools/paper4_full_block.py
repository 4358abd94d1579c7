"""ONE whole production block of the reference's benchmark shape (paper4: 84 x 84 output stamps, 1849 PSF groups, every stamp through the
Cholesky repair) through coadd_block, wall clock around all of it:  python tools/paper4_full_block.py > gpurun_out/paper4_block.json
(bench.py's paper4 leg times the first three passes and extrapolates; this is the number it extrapolates to)."""
import json, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from pyimcom_amd import psfs as psfmod
from pyimcom_amd._lib import Context
from pyimcom_amd.blockrun import coadd_block, memory_plan, plan_block
from pyimcom_amd.stamps import BlockTables

dev, n1P = "cuda:0", 84
ctx = Context(0)
cfg, inst, pool, psfs_b, target_b, groups, counts, img_all, yxco_all = bench.block_workload(dev, n1P, config="paper4")
ns, order = psfs_b.shape[-1], {k: q for q, k in enumerate(groups)}


def sample_groups(keys):
    idx = torch.tensor([order[k] for k in keys]).pin_memory().to(dev, non_blocking=True)
    return psfmod.sample_psf(img_all[idx].reshape(-1, ns + 16, ns + 16), ns, yxco_all[idx].reshape(-1, 2, ns, ns), psf_norm=True, ctx=ctx)


mplan = memory_plan(cfg, pool, n1P, cfg.n_expo, ctx=ctx)
tabs = BlockTables(groups, target_b, cfg.nfft, ctx=ctx, device=dev, group_count=counts, bulk_provider=sample_groups, cells=True,
                   capacity=mplan["capacity"], spec_capacity=mplan["spec_capacity"])
plan = plan_block(cfg, pool, tabs, n1P)
coadd_block(cfg, pool, tabs, n1P, cfg.n_expo, chunks=plan[:1], pad_sides=None)  # warm-up: buffers, workspace, kernels
torch.cuda.synchronize()
tabs.reset()
tel = bench.Telemetry(0).start()
ctx.profile_enable(True); ctx.profile_reset()
t0 = time.perf_counter()
maps = coadd_block(cfg, pool, tabs, n1P, cfg.n_expo, chunks=plan)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
fam = ("psf_sample", "psf_spectra", "psf_overlap", "select", "build_A", "build_B", "chol_gemm", "chol_diag", "eigen_repair", "solve_gemm", "finalize", "epilogue", "block_acc")
st = {f: round(ctx.profile_get(f)[0], 1) for f in fam}
out_map = maps.out_map.float().cpu().numpy()
ps = [float(t) for t in maps.pass_seconds]
if len(sys.argv) > 1:  # the plan's passes (for IMCOM_LMIN_DUMP=1 runs: which stamp each dumped eigenvalue belongs to)
    json.dump([[list(map(int, t)) for t in c] for c in plan], open(sys.argv[1], "w"))
print(json.dumps({"workload": "paper4 production block: 84 x 84 output stamps, n2 = 32, fade 3, INPAD 1.24, 6 exposures, 6 layers, a PSF group per 2 x 2 InStamps",
                  "stamps": n1P * n1P, "seconds": dt, "stamps_per_s": n1P * n1P / dt, "ms_per_stamp": dt * 1e3 / (n1P * n1P), "passes": len(plan),
                  "pass_sizes": sorted({len(c) for c in plan}), "pass_seconds": {"first": ps[0], "median": float(np.median(ps)), "max": max(ps), "last": ps[-1]},
                  "passes_halved": int(maps.passes_halved), "stamps_repaired": int(maps.info_nonzero), "tables": {"computed": int(tabs.computed_tables), "block_total": int(tabs.block_demand()), "arena": int(tabs.capacity)},
                  "stage_ms": {k: v for k, v in st.items() if v > 0}, "memory_plan": {k: mplan[k] for k in ("capacity", "spec_capacity", "stamps", "bytes_per_stamp")},
                  "out_map": {"shape": list(out_map.shape), "finite": bool(np.isfinite(out_map).all()), "rms": float(np.sqrt(np.mean(np.square(out_map))))},
                  "telemetry": tel.stop()}))
