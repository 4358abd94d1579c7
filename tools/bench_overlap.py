"""Would overlapping two half-batches pay?  One StampBatch of 256 cfg-2 stamps (the headline's step) against two of 128 on two contexts /
streams, their stages interleaved so that one half's A / B builds (gather-bound, matrix pipe idle) run beside the other half's
factorisation and solves (matrix-pipe-bound):   python tools/bench_overlap.py [steps]"""
import json
import sys
import time

sys.path.insert(0, ".")
import torch

from pyimcom_amd import synth
from pyimcom_amd._lib import Context
from pyimcom_amd.stamps import PSFGroupTables, StampBatch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = synth.CONFIGS["cfg2"]
stamps = sorted((synth.make_stamp(cfg, i) for i in range(256)), key=lambda st: -st.n)
psfs, target = synth.make_psfs(cfg, cfg.n_expo)


def rate(fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 256 * n / (time.perf_counter() - t0)


ctx = Context(0)
tabs = PSFGroupTables(psfs, target, cfg.nfft, ctx=ctx)
one = StampBatch(cfg, stamps, tabs, ctx=ctx)
r_one = rate(one.run, steps)
del one
torch.cuda.empty_cache()

ctxs = [ctx, Context(0)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
halves = []
for k in range(2):
    with torch.cuda.stream(streams[k]):
        t_ = tabs if k == 0 else PSFGroupTables(psfs, target, cfg.nfft, ctx=ctxs[k])
        halves.append(StampBatch(cfg, stamps[k::2], t_, ctx=ctxs[k]))
torch.cuda.synchronize()


def two():
    # stage by stage, alternating the halves: every call only queues work on its half's stream (solve_end waits for its half's flags)
    for k in range(2):
        with torch.cuda.stream(streams[k]):
            halves[k].build()
            halves[k].solve_begin()
    for k in range(2):
        with torch.cuda.stream(streams[k]):
            halves[k].coadd()
            halves[k].solve_end()


def two_staggered():
    # half 1's builds are queued after half 0's solve has been queued: builds of one half beside the solves of the other
    with torch.cuda.stream(streams[0]):
        halves[0].build()
        halves[0].solve_begin()
    with torch.cuda.stream(streams[1]):
        halves[1].build()
        halves[1].solve_begin()
    with torch.cuda.stream(streams[0]):
        halves[0].coadd()
        halves[0].solve_end()
    with torch.cuda.stream(streams[1]):
        halves[1].coadd()
        halves[1].solve_end()


def two_serial():
    for k in range(2):
        with torch.cuda.stream(streams[k]):
            halves[k].run()
        torch.cuda.synchronize()


print(json.dumps({"one_batch_of_256": r_one, "two_halves_interleaved": rate(two, steps), "two_halves_staggered": rate(two_staggered, steps),
                  "two_halves_one_after_the_other": rate(two_serial, steps)}))
