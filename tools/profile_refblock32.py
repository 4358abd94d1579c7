"""refblock.coadd_output_stamps on a block of several passes (duck-typed n1P x n1P block): wall time of a cold and of warm calls, GPU time of
the stages, cProfile of the device loop's thread for the last call.   python tools/profile_refblock32.py [n1P=32] [threads=16]"""
import cProfile, pstats, sys, time
sys.path.insert(0, ".")
import torch
from pyimcom_amd import synth
from pyimcom_amd._lib import default_context
from pyimcom_amd.refblock import coadd_output_stamps
n1P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cfg = synth.CONFIGS["cfg2"]
blk, psfgrp, _, _ = synth.duck_block(cfg, n1P, cfg.n_expo, seed=5)
ctx = default_context()
fams = ("psf_sample", "psf_spectra", "psf_overlap", "select", "build_A", "build_B", "chol_gemm", "chol_diag", "solve_gemm", "finalize", "epilogue", "block_acc")
for rep in range(3):
    ctx.profile_enable(True); ctx.profile_reset()
    pr = cProfile.Profile() if rep == 2 else None
    t = time.perf_counter()
    if pr: pr.enable()
    coadd_output_stamps(blk, psfgrp, ctx=ctx, host_threads=thr)
    torch.cuda.synchronize()
    if pr: pr.disable()
    dt = time.perf_counter() - t
    gpu = sum(ctx.profile_get(f)[0] for f in fams)
    ctx.profile_enable(False)
    print(f"call {rep}: wall {dt * 1e3:.1f} ms, GPU stages {gpu:.1f} ms, {n1P * n1P / dt:.1f} stamps/s", flush=True)
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
