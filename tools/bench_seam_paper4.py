"""The kernel-class seam on ONE production-shape stamp (paper4: N ~ 6.2k, m = 1444; 311 MB of A over PCIe), several calls in a row as the
reference's stamp loop makes them (neighbours: each call starts from the smallest eigenvalues of the calls before it):
    [IMCOM_LMIN_HINT=0] python tools/bench_seam_paper4.py [calls]"""
import json, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from pyimcom_amd import synth
from pyimcom_amd.lakernel import HipCholKernel
from pyimcom_amd.stamps import PSFGroupTables, StampBatch

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = synth.CONFIGS["paper4"]
stamps = [synth.make_stamp(cfg, i) for i in range(calls)]
psfs, target = synth.make_psfs(cfg, cfg.n_expo)
tabs = PSFGroupTables(psfs, target, cfg.nfft)
b = StampBatch(cfg, stamps, tabs)
b.build(); torch.cuda.synchronize()


class O:
    pass


def outst(s):
    n, m = int(b.n[s]), cfg.m
    o, o.blk = O(), O()
    o.blk.cfg = O()
    c = o.blk.cfg
    c.n_out, c.n2f, c.kappaC_arr, c.uctarget, c.sigmamax = 1, cfg.n2f, np.array(cfg.kappaC), cfg.uctarget, cfg.sigmamax
    o.sysmata = b.A[s, :n, :n].cpu().numpy().copy()
    o.mhalfb = np.ascontiguousarray(b.Bt[s, :n, :m].cpu().numpy().T)[None]
    o.outovlc, o.inpix_cumsum = np.array([tabs.C]), np.array([n])
    return o


os_ = [outst(s) for s in range(calls)]
ms, stages = [], []
fams = ("pack", "chol_gemm", "chol_diag", "eigen_repair", "solve_gemm", "finalize")
for o in os_:
    b.ctx.profile_enable(True)
    b.ctx.profile_reset()
    t0 = time.perf_counter()
    k = HipCholKernel(o, ctx=b.ctx)
    k()
    ms.append(round((time.perf_counter() - t0) * 1e3, 1))
    stages.append({f: round(b.ctx.profile_get(f)[0], 1) for f in fams})
    b.ctx.profile_enable(False)
print(json.dumps({"ms_per_call": ms, "info": [int(k.info[0])], "N": int(b.n[0]), "m": cfg.m, "stages_last": stages[-1], "stages_first": stages[0]}))
# the raw transfers of one call: A and -B/2 up from pageable numpy arrays, T down into a pinned array
A_h, B_h = os_[0].sysmata, os_[0].mhalfb
d = torch.empty(A_h.size, dtype=torch.float64, device="cuda:0")
torch.cuda.synchronize()
t0 = time.perf_counter(); d.copy_(torch.from_numpy(A_h).reshape(-1)); torch.cuda.synchronize(); tA = time.perf_counter() - t0
Ap = torch.from_numpy(A_h).pin_memory()
t0 = time.perf_counter(); d.copy_(Ap.reshape(-1), non_blocking=True); torch.cuda.synchronize(); tAp = time.perf_counter() - t0
print(json.dumps({"A_MB": A_h.nbytes / 1e6, "A_up_pageable_ms": tA * 1e3, "A_up_pinned_ms": tAp * 1e3, "B_MB": B_h.nbytes / 1e6}))
