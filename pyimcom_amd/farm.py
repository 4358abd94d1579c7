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
