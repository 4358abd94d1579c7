"""Pure MFMA loop in its diagnostic modes (IMCOM_MFMA_PROBE_MODE, read per launch): python tools/probe_mfma_modes.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pyimcom_amd._lib import default_context
ctx = default_context()
for mode, name in ((0, "constant operands"), (1, "pseudo-random operands in registers"), (2, "pseudo-random operands re-read from LDS per 8 MFMAs"),
                   (0, "constant again")):
    os.environ["IMCOM_MFMA_PROBE_MODE"] = str(mode)
    print(f"mode {mode} {name:52s}", [round(ctx.mfma_probe(400.0), 1) for _ in range(3)])
print("gemm probe 2304^2 x 8 (2592 tiles = 5.06 rounds of 512)", [round(ctx.gemm_probe(0, 2304, 2304, 2304, 8, 10), 1) for _ in range(2)])
print("gemm probe 2048^2 x 8 (2048 tiles = 4 rounds)", [round(ctx.gemm_probe(0, 2048, 2048, 2304, 8, 10), 1) for _ in range(2)])
for v, name in ((3, "k-major x k-major"), (5, "  without the LDS-DMA of later slices"), (6, "  without the per-slice wait + barrier"), (7, "  without both"), (8, "  DMA + barrier, no vmcnt wait")):
    print(f"gemm probe variant {v} {name:40s}", [round(ctx.gemm_probe(v, 2048, 2048, 2304, 8, 10), 1) for _ in range(2)])
