"""Runs one of the rate probes for a few seconds (for tools/poll_clocks.sh):  python tools/probe_loop.py mfma|gemm0|gemm1 [seconds]"""
import sys, time
from pyimcom_amd._lib import default_context
ctx = default_context()
which, secs = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
t0, vals = time.time(), []
while time.time() - t0 < secs:
    vals.append(ctx.mfma_probe(500.0) if which == "mfma" else ctx.gemm_probe(int(which[-1]), 2304, 2304, 2304, 8, 20))
print(which, "TFLOP/s", [round(v, 1) for v in vals])
