#!/bin/bash
# HBM traffic of the band reduction's kernels (two --pmc passes over ONE reduction of 8 matrices of 2944 rows): tools/pmc_band_reduce.sh
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export PYTHONPATH=$ROOT
O=$ROOT/gpurun_out/pmc_band
rm -rf $O; mkdir -p $O
cat > $O/run.py <<'PY'
import sys, numpy as np, torch, ctypes as C
from pyimcom_amd._lib import MEM_DEVICE, check, lib, default_context
b, ld, n = 8, 2944, 2938
g = torch.Generator(device="cuda").manual_seed(1)
X = torch.randn((b, ld, 400), dtype=torch.float64, device="cuda", generator=g)
A = X @ X.transpose(1, 2) / 400 + 0.05 * torch.eye(ld, dtype=torch.float64, device="cuda")
band = torch.zeros((b, 5, ld), dtype=torch.float64, device="cuda"); V = torch.zeros((b, ld, ld), dtype=torch.float64, device="cuda")
tau = torch.zeros((b, ld), dtype=torch.float64, device="cuda")
ns = np.full(b, n, dtype=np.int32)
ctx = default_context()
p = lambda t: C.c_void_p(t.data_ptr())
torch.cuda.synchronize()
check(lib.imcom_band_reduce(ctx.handle, b, ns.ctypes.data_as(C.c_void_p), ld, p(A), p(band), p(V), p(tau), MEM_DEVICE))
torch.cuda.synchronize()
print("done", flush=True)
PY
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 280 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "symv4|band_apply|band_step|syr2k" -d $O/$c -- python3 $O/run.py > $O/$c.log 2>&1 || { echo "pass $c failed"; tail -3 $O/$c.log; exit 1; }
  echo "pass $c done"
done
cd $ROOT
F=$(find $O/FETCH_SIZE -name '*.db' | head -1); W=$(find $O/WRITE_SIZE -name '*.db' | head -1)
python tools/pmc_traffic.py "$F" "$W" $O/pmc_band.json "band_reduce n=2938" 8 "tools/pmc_band_reduce.sh: one reduction of 8 matrices"
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
