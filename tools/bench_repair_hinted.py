"""The Cholesky repair on paper4 stamps as a block's passes run it: a blind pass (expected repair, no hint), then hinted passes from the blind
pass's record:  [IMCOM_LMIN_SKINNY=0|1] [IMCOM_LMIN_DEBUG=1] python tools/bench_repair_hinted.py [batch] [reps] [check]
One JSON line: ms per stamp (whole solve / eigen_repair / chol_gemm / solve_gemm) of the hinted passes, the smallest eigenvalues' range
(imcom_ctx_last_repair), digests of T and the maps; `check`: torch.linalg.eigvalsh of every stamp's A (the tool's cross-check) beside them."""
import json, os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from pyimcom_amd import synth
from pyimcom_amd.stamps import PSFGroupTables, StampBatch

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = synth.CONFIGS["paper4"]
stamps = [synth.make_stamp(cfg, i) for i in range(nb)]
psfs, target = synth.make_psfs(cfg, cfg.n_expo)
b = StampBatch(cfg, stamps, PSFGroupTables(psfs, target, cfg.nfft))
b.build(); torch.cuda.synchronize()
t0 = time.perf_counter()
b.solve_begin(expect_repair=True); b.solve_end(); torch.cuda.synchronize()
blind_ms = (time.perf_counter() - t0) * 1e3 / nb
hint = b.repair_absmax
cnt, lo, hi = b.ctx.last_repair()
b.ctx.profile_enable(True); b.ctx.profile_reset()
t0 = time.perf_counter()
for _ in range(reps):
    b.solve_begin(expect_repair=True, repair_hint=hint); b.solve_end()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
fam = {f: b.ctx.profile_get(f)[0] / reps / nb for f in ("eigen_repair", "chol_gemm", "solve_gemm")}
cnt2, lo2, hi2 = b.ctx.last_repair()
T = b.Tt_o[0]
out = {"batch": nb, "blind_ms_per_stamp": round(blind_ms, 3), "hinted_ms_per_stamp": round(dt * 1e3 / nb, 3), **{k: round(v, 3) for k, v in fam.items()},
       "repaired": int((b.info_o[0] != 0).sum()), "hint": hint, "w0_blind": [cnt, lo, hi], "w0_hinted": [cnt2, lo2, hi2],
       "T_digest": [float(T.double().abs().sum()), float(T.double().reshape(-1)[::9973].sum())],
       "maps_digest": [float(b.UC_o[0].double().sum()), float(b.Sigma_o[0].double().sum()), float(b.kappa_o[0].double().sum())]}
if len(sys.argv) > 3:
    w = []
    for s in range(nb):
        n = int(b.n[s])
        w.append(float(torch.linalg.eigvalsh(b.A[s, :n, :n])[0]))
    out["eigvalsh_w0"] = [min(w), max(w)]
    out["rel_dev_min_max"] = [abs(lo2 - min(w)) / abs(min(w)), abs(hi2 - max(w)) / abs(max(w))]
print(json.dumps(out))
