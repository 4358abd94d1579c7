"""Block farming over the GPUs of a node: one process per GPU, static partition, no data-path collective.

The reference's only parallelism is one OS process per block (docs/run_README.rst:81-100,
examples/multiblock_norep.pl:42-66); blocks share read-only inputs and write separate outputs, so ranks never
exchange data.  torch.distributed is used by callers only for start/end barriers.
"""

from typing import List, Sequence


def estimate_cost(n_pixels: float, m: int, nv: int = 1) -> float:
    """Relative cost of one stamp: factor N^3/3 + solve 2 N^2 m per kappa node."""
    return nv * (n_pixels**3 / 3.0 + 2.0 * n_pixels**2 * m)


def partition(costs: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of units (blocks) to `world` ranks.

    Deterministic (ties by index), so every rank computes the same partition without communicating."""
    if world < 1:
        raise ValueError("world must be >= 1")
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += costs[i]
    return [sorted(x) for x in out]


def my_units(costs: Sequence[float], rank: int, world: int) -> List[int]:
    return partition(costs, world)[rank]


def batches(units: Sequence[int], batch: int) -> List[List[int]]:
    """Split a rank's units into device batches (the last one may be short)."""
    return [list(units[i : i + batch]) for i in range(0, len(units), batch)]


# ---------------------------------------------------------------------------------------------------------------
# The driver: blocks -> ranks -> coadd_block -> one output file per block.
#
# The reference coadds a mosaic by starting one OS process per block, skipping blocks whose output file is already
# there (examples/multiblock_norep.pl:25-27, 42-66; docs/run_README.rst:81-100).  Here one process per GPU walks its
# share of the block list; a block's inputs come from a `make_block(b)` callable, so the same driver serves the
# synthetic mosaics of the benchmarks and a maintainer's real blocks (make_block = what Block.__init__ prepares,
# see pyimcom_amd.refblock).  Ranks never exchange data.
def block_path(outdir, b):
    import os

    return os.path.join(outdir, f"block_{int(b):04d}.npz")


def write_block(path, maps, meta=None):
    """One .npz per block, written under a temporary name and renamed, so that a killed run never leaves a file that
    a restart would mistake for a finished block."""
    import os

    import numpy as np

    out = {"out_map": maps.out_map.cpu().numpy(), "T_weightmap": maps.T_weightmap.cpu().numpy()}
    out.update({k: v.cpu().numpy() for k, v in maps.maps.items()})
    for k, v in (meta or {}).items():
        out["meta_" + k] = np.asarray(v)
    tmp = path + f".tmp{os.getpid()}.npz"
    np.savez(tmp, **out)
    os.replace(tmp, path)


def run(blocks, costs, make_block, outdir, rank=0, world=1, batch=None, device=None, restart=True, log=print, coadd=None):
    """Coadd this rank's share of `blocks` (ids) and write block_<id>.npz files into `outdir`.

    costs[k]: relative cost of blocks[k] (estimate_cost summed over its stamps) for the static LPT partition;
    make_block(b) -> dict(cfg=, pool=, tables=, n1P=, n_expo=, [pad_sides=, postage_pad=, meta=]) -- the arguments of
    pyimcom_amd.blockrun.coadd_block -- called on the rank that owns b, right before b is coadded;
    restart: skip blocks whose file exists (multiblock_norep.pl:25-27).  `coadd` replaces
    pyimcom_amd.blockrun.coadd_block (the host-logic tests run the driver without a GPU that way).
    Returns the list of block ids done here."""
    import os
    import time

    import torch

    os.makedirs(outdir, exist_ok=True)
    on_gpu = coadd is None
    if on_gpu:
        from .blockrun import coadd_block as coadd

        dev = device or f"cuda:{rank % max(torch.cuda.device_count(), 1)}"
        torch.cuda.set_device(dev)
    mine = [blocks[k] for k in my_units(costs, rank, world)]
    done = []
    for b in mine:
        path = block_path(outdir, b)
        if restart and os.path.exists(path):
            log(f"[farm rank {rank}] block {b}: {path} exists, skipped")
            continue
        t0 = time.perf_counter()
        spec = make_block(b)
        maps = coadd(spec["cfg"], spec["pool"], spec["tables"], spec["n1P"], spec["n_expo"], batch=batch,
                     pad_sides=spec.get("pad_sides", ""), postage_pad=spec.get("postage_pad", 0))
        if on_gpu:
            torch.cuda.synchronize()
        write_block(path, maps, spec.get("meta"))
        done.append(b)
        log(f"[farm rank {rank}] block {b}: {spec['n1P'] ** 2} stamps, {spec['n_expo']} exposures, {time.perf_counter() - t0:.2f} s -> {path}")
    return done


def synthetic_mosaic(config="cfg4", nblock=4, n1P=2, seed=4, psf_groups=False):
    """The cfg-4 workload of BASELINE.json as a block list: nblock x nblock blocks of n1P x n1P output stamps, every
    block with its own exposure depth (uniform in the configuration's range), lattices and data.  Returns
    (blocks, costs, make_block)."""
    import dataclasses

    import numpy as np

    from . import synth

    base = synth.CONFIGS[config]
    rng = np.random.default_rng(seed)
    lo, hi = base.n_expo if isinstance(base.n_expo, tuple) else (base.n_expo, base.n_expo)
    depth = rng.integers(lo, hi + 1, nblock * nblock)
    blocks = list(range(nblock * nblock))
    p = synth.NATIVE_ARCSEC / base.dtheta_as
    n_pix = lambda e: e * (base.n2 + 2 * base.rho) ** 2 / p**2  # noqa: E731  input pixels of one stamp, roughly
    costs = [n1P * n1P * estimate_cost(n_pix(int(e)), base.m, len(base.kappaC)) for e in depth]

    def make_block(b, device="cuda:0"):
        import torch

        from ._lib import default_context
        from .select import InStampPool
        from .stamps import BlockTables, PSFGroupTables

        ctx = default_context(torch.device(device).index or 0)  # one context per GPU
        E = int(depth[b])
        cfg = dataclasses.replace(base, n_expo=E, name=f"{base.name}_b{b}")
        inst = synth.make_instamps(cfg, n1P, E, np.random.default_rng([seed, b]))
        psfs, target = synth.make_psfs(cfg, E, seed=20260723 + b)
        if psf_groups:
            ng = (n1P + 3) // 2
            tables = BlockTables({(gj, gi): psfs for gj in range(ng) for gi in range(ng)}, target, cfg.nfft, capacity=2048, ctx=ctx, device=device, cells=True)
        else:
            tables = PSFGroupTables(psfs, target, cfg.nfft, ctx=ctx, device=device)
        return dict(cfg=cfg, pool=InStampPool(inst, cfg.n_inframe, device=device), tables=tables, n1P=n1P, n_expo=E,
                    meta=dict(block=b, n_expo=E, config=config))

    return blocks, costs, make_block


def main(argv=None):
    """python -m pyimcom_amd.farm --out DIR [--config cfg4 --mosaic 4 --n1P 2]: coadd a synthetic mosaic, this process
    taking the blocks of rank RANK of WORLD_SIZE (environment, as torch.distributed.run sets them; default 0 of 1).
    One process per GPU (LOCAL_RANK); --shared-gpu puts every rank on cuda:0 (a rehearsal on a one-GPU box)."""
    import argparse
    import os

    ap = argparse.ArgumentParser(prog="python -m pyimcom_amd.farm", description=main.__doc__)
    ap.add_argument("--out", required=True)
    ap.add_argument("--config", default="cfg4")
    ap.add_argument("--mosaic", type=int, default=4, help="blocks per side")
    ap.add_argument("--n1P", type=int, default=2, help="output stamps per block side")
    ap.add_argument("--batch", type=int, default=None, help="stamps per pass (default: blockrun.choose_batch)")
    ap.add_argument("--seed", type=int, default=4)
    ap.add_argument("--psf-groups", action="store_true", help="a PSF group per 2x2 InStamps (BlockTables) instead of one per block")
    ap.add_argument("--no-restart", action="store_true", help="recompute blocks whose output exists")
    ap.add_argument("--shared-gpu", action="store_true")
    a = ap.parse_args(argv)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if a.shared_gpu else int(os.environ.get("LOCAL_RANK", str(rank)))
    blocks, costs, make_block = synthetic_mosaic(a.config, a.mosaic, a.n1P, a.seed, a.psf_groups)
    dev = f"cuda:{local}"
    done = run(blocks, costs, lambda b: make_block(b, dev), a.out, rank, world, a.batch, device=dev, restart=not a.no_restart)
    print(f"[farm rank {rank}/{world}] done: {done}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
