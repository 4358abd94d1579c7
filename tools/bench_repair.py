"""The Cholesky repair of the paper4 shape alone (api.hip lambda_min_subspace):  [IMCOM_LMIN_DEBUG=1] python tools/bench_repair.py [batch] [reps] [check]
One JSON line: ms per stamp of a whole run() and of the eigen_repair family; with `check` the smallest eigenvalue the run used (info = 1 stamps:
shift = kappa + |w0| + 1e-16 read back through T is not possible, so the check is LAPACK's w[0] of two stamps' A against IMCOM_LMIN=eigh's path
-- run the script twice, once per setting, and compare `T_digest`)."""
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
b.run(); torch.cuda.synchronize()
b.ctx.profile_enable(os.environ.get('BENCH_NOPROF') != '1'); b.ctx.profile_reset()
t0 = time.perf_counter()
for _ in range(reps):
    res = b.run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
fam = {f: b.ctx.profile_get(f)[0] / reps / nb for f in ("eigen_repair", "chol_gemm", "solve_gemm", "build_A")}
T0 = res.T(0).float().cpu().numpy()
out = {"batch": nb, "ms_per_stamp": round(dt * 1e3 / nb, 3), **{k: round(v, 3) for k, v in fam.items()}, "repaired": int((b.info != 0).sum()),
       "T_digest": [float(np.abs(T0).sum()), float(T0.ravel()[::9973].sum())]}
if len(sys.argv) > 3:
    n = int(b.n[0])
    w0 = np.linalg.eigvalsh(b.A[0, :n, :n].cpu().numpy())[0]
    out["lapack_w0_stamp0"] = float(w0)
print(json.dumps(out))
