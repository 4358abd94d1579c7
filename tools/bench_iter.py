"""The Iterative kernel at the reference's default configuration (synth.CONFIGS["iter_default"]) alone: resident batches, stage times,
the blocked CG's flops / bytes, and (argument 3 != 0) one stamp against the oracle.
    python tools/bench_iter.py [batch] [steps] [check]"""
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from pyimcom_amd import synth
from pyimcom_amd._lib import Context
from pyimcom_amd.stamps import PSFGroupTables, StampBatch

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
check = int(sys.argv[3]) if len(sys.argv) > 3 else 1
name = sys.argv[4] if len(sys.argv) > 4 else "iter_default"
cfg = synth.CONFIGS[name]
ctx = Context(0)
stamps = [synth.make_stamp(cfg, i) for i in range(nb)]
psfs, target = synth.make_psfs(cfg, cfg.n_expo)
tables = PSFGroupTables(psfs, target, cfg.nfft, ctx=ctx, device="cuda:0")
b = StampBatch(cfg, stamps, tables, ctx=ctx, device="cuda:0")
b.run()
torch.cuda.synchronize()
fams = ("build_A", "build_B", "iter_gather", "iter_cg", "epilogue", "finalize")
ctx.profile_enable(True)
ctx.profile_reset()
t0 = time.perf_counter()
for _ in range(steps):
    b.run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
st = {f: ctx.profile_get(f)[0] / steps for f in fams}
ctx.profile_enable(False)
its = b.iter_stats
cg_s = st["iter_cg"] * 1e-3
out = {"config": name, "batch": nb, "ms_per_stamp": dt / nb * 1e3, "stamps_per_s": nb / dt, "stage_ms_per_step": st, "N_mean": float(b.n.mean()),
       "iter": its, "cg_TFLOPs": its["flops"] / cg_s / 1e12 if cg_s else None, "cg_GBs": its["bytes"] / cg_s / 1e9 if cg_s else None,
       "steps_per_patch": its["patch_steps"] / max(its["patches"], 1)}
print(json.dumps(out))
if check:
    from oracle import oracle as orc

    g, tabs_ref, C_ref = orc.stamp_tables(cfg, psfs, target)
    E = cfg.n_expo
    tri = lambda i, j: (2 * E - i + 1) * i // 2 + j - i
    tab = np.array([[tri(a_, b_) if a_ <= b_ else (tri(b_, a_) | (1 << 30)) for b_ in range(E)] for a_ in range(E)], dtype=np.int32)
    pen = np.array([[-cfg.flat_penalty / E + (cfg.flat_penalty if a_ == b_ else 0.0) for b_ in range(E)] for a_ in range(E)])
    io = np.arange(E) + E * (E + 1) // 2
    t0 = time.perf_counter()
    ref = orc.stamp_full(cfg, g, tabs_ref, float(C_ref[0]), stamps[0], tab, pen, io)
    t_or = time.perf_counter() - t0
    # the oracle's CG again with its step counts (on the oracle's own A and -B/2)
    g1 = np.arange(cfg.n2f, dtype=np.float64)
    oy, ox = np.repeat(stamps[0].out_y0 + g1, cfg.n2f), np.tile(stamps[0].out_x0 + g1, cfg.n2f)
    osteps = []
    orc.iter_kernel(ref["A"], np.ascontiguousarray(ref["Bt"].T), float(C_ref[0]), np.array(cfg.kappaC), cfg.uctarget, cfg.sigmamax, oy, ox, stamps[0].y, stamps[0].x,
                    cfg.rho, cfg.iter_rtol, cfg.iter_max, steps=osteps)
    osteps = np.array(osteps)
    res = b.result()
    n0 = stamps[0].n
    Tg = res.T(0).cpu().numpy()
    Tr = ref["T"]
    rel = float(np.abs(Tg - Tr).max() / np.abs(Tr).max())
    per_pix = np.abs(Tg - Tr).max(axis=1) / np.abs(Tr).max(axis=1)
    _, steps_px = ctx.iter_stats(nb * cfg.m)
    img = res.outimage[0].cpu().numpy()
    gs = steps_px[: cfg.m]
    same = gs == osteps
    print(json.dumps({"steps_equal_share": float(same.mean()), "T_rel_max_where_steps_equal": float(per_pix[same].max()), "T_rel_median_where_differ": float(np.median(per_pix[~same])) if (~same).any() else None,
                      "step_diff_hist": np.bincount(np.abs(gs - osteps)).tolist(), "oracle_steps_hist": np.bincount(osteps, minlength=31).tolist()}))
    Ag = b.A[0, :n0, :n0].cpu().numpy()
    print(json.dumps({"A_rel": float(np.abs(Ag - ref["A"]).max() / np.abs(ref["A"]).max()), "A_eig_min_max": [float(v) for v in np.linalg.eigvalsh(ref["A"])[[0, -1]]]}))
    print(json.dumps({"oracle_s": t_or, "T_rel_max": rel, "T_rel_median_pixel": float(np.median(per_pix)), "T_rel_p99": float(np.quantile(per_pix, 0.99)),
                      "steps_hist": np.bincount(steps_px[: cfg.m], minlength=31).tolist()[-6:], "UC_max_diff": float(np.abs(res.UC[0].cpu().numpy() - ref["UC"]).max()),
                      "Sigma_rel": float(np.abs(res.Sigma[0].cpu().numpy() - ref["Sigma"]).max() / np.abs(ref["Sigma"]).max()),
                      "img_rel": float(np.abs(img - ref["outimage"]).max() / np.abs(ref["outimage"]).max()), "maxT": float(np.abs(Tr).max())}))
