"""Block farming: the static partition, and a world_size-2 gloo run on CPU showing that ranks cover the
blocks disjointly with no data-path collective (only the start/end barrier and a checksum gather)."""

import os

import numpy as np
import pytest

from pyimcom_amd import farm


def test_partition_properties():
    rng = np.random.default_rng(1)
    costs = rng.uniform(1, 10, 16).tolist()  # cfg-4: 4x4 blocks, variable depth
    for world in (1, 2, 4, 8):
        parts = farm.partition(costs, world)
        assert sorted(sum(parts, [])) == list(range(16))
        loads = [sum(costs[i] for i in p) for p in parts]
        assert max(loads) <= sum(costs) / world + max(costs)  # LPT bound
    assert farm.partition(costs, 8) == farm.partition(costs, 8)  # deterministic
    assert farm.batches(list(range(10)), 4) == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]]
    assert farm.estimate_cost(2200, 2304) > farm.estimate_cost(1900, 1024)


def _worker(rank, world, port, out):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pyimcom_amd import synth

    cfg = synth.CONFIGS["cfg4"]
    # 16 blocks of one stamp each (synthetic): cost from the stamp's pixel count
    stamps = [synth.make_stamp(cfg, b) for b in range(16)]
    costs = [farm.estimate_cost(s.n, cfg.m) for s in stamps]
    mine = farm.my_units(costs, rank, world)
    dist.barrier()
    # "coadd" = a per-block checksum of the inputs; nothing is exchanged while working
    res = {b: float(stamps[b].x.sum() + stamps[b].y.sum()) for b in mine}
    dist.barrier()
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        out.put(gathered)
    dist.destroy_process_group()


def test_gloo_world2_cover():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    keys = sorted(k for g in gathered for k in g)
    assert keys == list(range(16)) and len(gathered[0]) + len(gathered[1]) == 16
    assert not set(gathered[0]) & set(gathered[1])
