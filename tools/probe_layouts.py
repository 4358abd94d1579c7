"""The tile engine's k loop in its four operand layouts (imcom_ctx_gemm_probe variants 0, 2, 3, 4), TFLOP/s:
   python tools/probe_layouts.py [K=2304] [batch=8]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pyimcom_amd._lib import default_context
ctx = default_context()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2304
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
print("pure MFMA", round(ctx.mfma_probe(300.0), 1))
for v, name in ((0, "A row-major, B k-major (forward solve)"), (2, "A row-major, B row-major (Cholesky update)"),
                (3, "A k-major, B k-major (backward solve)"), (4, "A k-major, B row-major"), (0, "A row-major, B k-major again")):
    print(f"variant {v} {name:46s}", [round(ctx.gemm_probe(v, 2304, 2304, K, B, 10), 1) for _ in range(3)])
