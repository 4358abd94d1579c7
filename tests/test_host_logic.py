"""Host-side logic that needs no GPU: stamp neighbourhoods and pivots (coadd.py:853, 918-919), the visiting order of
the pixel partition (coadd.py:329-336), the table sets a stamp needs, and the oracle's own selection / block helpers."""

import os

import numpy as np
import pytest


def test_stamp_neighbours_and_pivots():
    from pyimcom_amd.blockrun import stamp_neighbours

    n2, nst = 8, 6
    ids, pvx, pvy = stamp_neighbours(2, 3, n2, nst)
    assert ids.tolist() == [1 * nst + 2, 1 * nst + 3, 1 * nst + 4, 2 * nst + 2, 2 * nst + 3, 2 * nst + 4, 3 * nst + 2, 3 * nst + 3, 3 * nst + 4]
    left, bottom = (3 - 1) * n2, (2 - 1) * n2
    assert pvx[0] == left - 0.5 and np.isnan(pvx[1]) and pvx[2] == left + n2 - 1 + 0.5
    assert pvy[0] == bottom - 0.5 and np.isnan(pvy[4]) and pvy[8] == bottom + n2 - 1 + 0.5
    assert np.isnan(pvx[4]) and np.isnan(pvy[4])  # the centre InStamp is taken whole
    ids, _, _ = stamp_neighbours(0, 0, n2, nst)  # outside the block: absent neighbours are -1
    assert ids.tolist() == [-1, -1, -1, -1, 0, 1, -1, nst, nst + 1]


def test_neighbour_tables_of_a_batch_equal_the_per_stamp_ones():
    """blockrun._neighbours_of builds the neighbour ids and pivots of a whole batch by broadcasting; it must give exactly what
    stamp_neighbours gives stamp by stamp (block corners, edges and stamps outside the block included)."""
    from pyimcom_amd.blockrun import _neighbours_of, stamp_neighbours

    n2, nst = 48, 7
    chunk = [(j, i) for j in range(0, nst) for i in range(0, nst)]
    for (j, i), got in zip(chunk, _neighbours_of(chunk, n2, nst)):
        for a, b in zip(stamp_neighbours(j, i, n2, nst), got):
            assert a.dtype == b.dtype and np.array_equal(a, b, equal_nan=True), (j, i)


def test_block_table_pair_lists_and_windows():
    """Storage order of the PSF pairs of a table set (triangle for a group's self overlap, psfutil.py:1175), and the window of a
    cross table of two grid-cell groups: the half plane(s) the separations r(g1) - r(g2) can reach, ten-tap stencils included."""
    from pyimcom_amd.stamps import BlockTables

    assert BlockTables._local_pairs("self", 3, 0).tolist() == [[0, 0], [0, 1], [0, 2], [1, 1], [1, 2], [2, 2]]
    assert BlockTables._local_pairs("io", 2, 2).tolist() == [[0, 0], [1, 0], [0, 1], [1, 1]]
    assert BlockTables._local_pairs("cross", 2, 3).tolist() == [[i, j] for i in range(2) for j in range(3)]
    assert BlockTables._local_pairs("cross", 2, 3) is BlockTables._local_pairs("cross", 2, 3)
    ns, nc = 383, 191
    lo, hi = (0, nc + 8), (nc - 8, ns)
    # a separation d < 0 lands in cell floor(d + nc + 6) <= nc + 5 of the bordered table, its taps reach table index nc + 10 =
    # window index nc + 4 < nc + 8; d > 0: cell >= nc + 6, taps from table index nc + 2 = window index nc - 4 >= nc - 8
    assert lo[1] > nc + 4 and hi[0] <= nc - 4


def test_visiting_order_matches_the_reference_loops():
    from pyimcom_amd.select import visiting_order

    sp_arr = np.linspace(0, 12, 4, dtype=np.uint16)  # 3 x 3 cells of 4 pixels
    rel = np.array([[1, 0, 1], [0, 0, 0], [0, 1, 0]], dtype=bool)
    ys, xs = visiting_order(rel, sp_arr)
    ref = [(b + j, l + i) for (jc, ic) in [(0, 0), (0, 2), (2, 1)] for b, l in [(sp_arr[jc], sp_arr[ic])] for j in range(4) for i in range(4)]
    assert list(zip(ys.tolist(), xs.tolist())) == ref and ys.dtype == np.uint16


def test_block_table_keys():
    from pyimcom_amd.stamps import BlockTables

    keys = BlockTables.keys_for([(1, 1), (0, 1), (1, 1), (0, 0)])
    assert keys[:3] == [("self", (0, 0)), ("self", (0, 1)), ("self", (1, 1))]
    assert ("cross", (0, 0), (1, 1)) in keys and ("cross", (1, 1), (0, 0)) not in keys and len(keys) == 3 + 3 + 3


def test_oracle_selection_and_block_helpers():
    from oracle import oracle as orc

    x = np.array([0.0, 1.0, 2.0, 3.0])
    y = np.zeros(4)
    assert orc.select_pixels(x, y, (None, None), 1.0) is None
    assert orc.select_pixels(x, y, (0.0, None), 10.0) is None  # everything selected -> None, as the reference returns
    assert orc.select_pixels(x, y, (0.0, None), 2.0).tolist() == [0, 1]  # strict '<': the pixel at distance 2 is out
    inst = (x, y, np.arange(8, dtype=np.float32).reshape(2, 4), np.array([0, 3, 4]))
    xs, ys, d, e, cum = orc.process_input_stamps([None] * 4 + [inst] + [None] * 4, [(None, None)] * 9, 1.0)
    assert xs.tolist() == x.tolist() and e.tolist() == [0, 0, 0, 1] and cum.tolist() == [0, 0, 0, 0, 0, 4, 4, 4, 4, 4]
    a = np.ones((1, 12, 12), np.float32)
    orc.trapezoid(a, 2)
    b = a.copy()
    orc.trapezoid_recover(b, 2)
    assert np.allclose(b, 1.0) and a[0, 0, 0] < 0.05
    assert orc.compress_map(np.array([1.0, 1e-40, 10.0], np.float32), -5000, np.uint16).tolist() == [0, 65535, 0]


def _partition_inputs(g):
    """Inputs of the partition in the reference's visiting order, from tests/golden/partition.npz."""
    from pyimcom_amd.select import visiting_order

    in_y, in_x = visiting_order(g["relevant"], g["sp_arr"])
    out = np.stack([in_x, in_y], axis=1).astype(np.float64) @ g["M"].T + g["t0"]  # as the generator's affine map, row by row
    mask = g["mask"][in_y, in_x]
    return in_y, in_x, out[:, 0], out[:, 1], mask


def test_partition_golden(golden):
    """The oracle's partition loop and the host's visiting order against the reference's own binning statement
    (coadd.py:329-358, executed by tests/golden/make_golden_partition.py): every array bit for bit."""
    from oracle import oracle as orc

    g = golden("partition")
    in_y, in_x, ox, oy, mask = _partition_inputs(g)
    got = orc.partition_pixels(ox, oy, in_x, in_y, mask, g["use_instamps"], int(g["n2"]), int(g["n1P"]), int(g["npixmax"]))
    for a, name in zip(got, ("y_idx", "x_idx", "y_val", "x_val", "pix_count")):
        assert a.dtype == g[name].dtype and np.array_equal(a, g[name]), name


class _ArenaStandIn:
    """What blockrun.plan_batches reads from a BlockTables: group sizes and the arena's capacity (no GPU needed)."""

    def __init__(self, n, capacity, n_out=1):
        self.n_max, self.n_out, self.capacity = n, n_out, capacity

    def demand(self, keys):
        n, o = self.n_max, self.n_out
        return sum(n * (n + 1) // 2 if k[0] == "self" else o * n if k[0] == "io" else n * n for k in dict.fromkeys(keys))


def test_plan_batches_tiles_cover_the_block_and_fit_the_arena():
    """The batches of a block with a PSF group per 2 x 2 InStamps (SysMatA.ji_st2psf, psfutil.py:1803-1824) are tiles of
    2 x 2-stamp cells: every stamp exactly once, no batch above the stamp limit, no batch whose table sets exceed the arena --
    at the reference's production geometries (n1P = 84 / 52: configs/paper4_configs/H158_Chol_benchmark.json:28-34,
    paper3_configs/H158_Chol_config.json:27-33) and exposure depths 6 - 10, where a row-by-row walk of 256 stamps needs more
    tables than a 2-D tile (the advisor's case: n1P = 16 with 8 and 10 PSFs per group against an arena of 13 500)."""
    from pyimcom_amd.blockrun import chunk_keys, plan_batches

    for n1P, n, capacity, cap in ((16, 6, 13500, 256), (16, 8, 13500, 256), (16, 10, 13500, 256), (48, 8, 30000, 256), (48, 10, 40000, 256),
                                  (84, 6, 13500, 256), (84, 10, 40000, 256), (52, 6, 6000, 256), (7, 6, 100000, 256), (84, 6, 10**6, 100)):
        nst = n1P + 2
        arena = _ArenaStandIn(n, capacity)
        todo = [(j, i) for j in range(1, n1P + 1) for i in range(1, n1P + 1)]
        chunks = plan_batches(todo, nst, cap, arena)
        assert sorted(t for c in chunks for t in c) == todo, (n1P, n)
        assert max(len(c) for c in chunks) <= cap
        dem = [arena.demand(chunk_keys(c, nst)) for c in chunks]
        assert max(dem) <= capacity, (n1P, n, max(dem))
        # the four stamps of a cell stay in one batch (they share their pair maps) unless a cell alone had to be split
        if min(len(c) for c in chunks) >= 4 and n1P % 2 == 0:
            where = {t: q for q, c in enumerate(chunks) for t in c}
            assert all(len({where[(j + dj, i + di)] for dj in (0, 1) for di in (0, 1)}) == 1 for j in range(1, n1P, 2) for i in range(1, n1P, 2))
        if n1P == 84 and n == 6 and capacity == 13500:
            rounds = sum(-(-len(c) * 18 // 512) for c in chunks)
            assert rounds <= 1.02 * -(-n1P * n1P * 18 // 512)  # whole rounds of workgroups: within 2 % of the block's minimum
            rows = [arena.demand(chunk_keys(todo[c0 : c0 + 256], nst)) for c0 in range(0, len(todo), 256)]  # a row-by-row walk of the block
            assert sum(dem) < 0.75 * sum(rows) and max(rows) > capacity
    # an arbitrary subset (the reference's stoptile-style partial runs) and the single-group case
    sub = [(j, i) for j in range(3, 30, 2) for i in range(5, 60, 3)]
    chunks = plan_batches(sub, 86, 64, _ArenaStandIn(6, 5000))
    assert sorted(t for c in chunks for t in c) == sorted(sub) and max(len(c) for c in chunks) <= 64
    assert plan_batches(sub, 86, 100, None) == [sub[c0 : c0 + 100] for c0 in range(0, len(sub), 100)]
    assert plan_batches([], 86, 100, None) == []


def test_choose_batch_is_kernel_aware_and_never_zero():
    from pyimcom_amd.blockrun import choose_batch, fill_of, max_stamps, pass_bytes

    assert choose_batch(0, 2304, 2304) == 1  # an empty block must not produce a zero step (coadd_block returns before using it)
    free = 100 << 30
    chol = choose_batch(2304, 2304, 2304, 1, free)
    assert chol == 256
    it = choose_batch(2304, 2304, 2304, 1, free, kernel="Iterative")
    # (the patch matrices of the blocked CG go through a fixed 8 GiB share of workspace since round 6 -- until then 0.4 GB per stamp)
    assert it <= chol and pass_bytes(it, 2304, 2304, 1, "Iterative") <= fill_of() * free
    assert 1 <= choose_batch(2304, 2304, 2304, 1, 30 << 30, kernel="Iterative") < 256
    eig = choose_batch(2304, 2944, 2304, 1, free, kernel="Eigen")
    assert pass_bytes(eig, 2944, 2304, 1, "Eigen") <= fill_of() * free
    # the exact bound: the largest pass that fits, by the library's own workspace arithmetic; one more stamp does not fit
    for kernel, ldn, avail in (("Cholesky", 6272, 120 << 30), ("Eigen", 3072, 60 << 30), ("Cholesky", 2304, 20 << 30)):
        k = max_stamps(avail, ldn, 1536, 1, kernel, nv=1, n_inframe=6)
        assert 1 <= k < 256 and pass_bytes(k, ldn, 1536, 1, kernel, 2, 1, 6) <= avail < pass_bytes(k + 1, ldn, 1536, 1, kernel, 2, 1, 6), (kernel, k)
    # three kappa nodes take a Y per node: fewer stamps fit
    assert max_stamps(60 << 30, 2304, 2304, 1, "Cholesky", nv=3) < max_stamps(60 << 30, 2304, 2304, 1, "Cholesky", nv=1)


def test_reference_stamp_order_and_host_ahead():
    """Host mirrors added in round 4: the visiting order of the reference's stamp loop (cells of 2 x 2 from the window's origin,
    coadd.py:2056-2064) against the fixture made by executing that loop, and refblock's _HostAhead (groups' host halves prepared
    on worker threads in the plan's order, on-demand requests served by the same workers, a bounded number held ready)."""
    import threading
    import time

    from pyimcom_amd.blockrun import reference_stamp_order
    from pyimcom_amd.refblock import _HostAhead

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "block_loop.npz"))
    for case in ("whole", "inner", "stop"):
        lo_j, hi_j, lo_i, hi_i = (int(v) for v in g[f"{case}_window"])
        assert np.array_equal(np.array(reference_stamp_order(lo_j, hi_j, lo_i, hi_i, int(g[f"{case}_nrun"]))), g[f"{case}_visited"])
    with pytest.raises(ValueError, match="Size must be even"):
        reference_stamp_order(1, 3, 1, 4)
    assert oracle_order(1, 4, 1, 2) == reference_stamp_order(1, 4, 1, 2)

    seen, lock = [], threading.Lock()

    def work(key):
        with lock:
            seen.append((key, threading.current_thread().name))
        time.sleep(0.01)
        return key * 2

    for threads in (1, 4):
        seen.clear()
        ah = _HostAhead(work, threads=threads, ahead=3)
        ah.schedule([1, 2, 3, 4, 5, 2])
        time.sleep(0.1)
        assert len(seen) == 3  # no more than `ahead` groups are prepared before anything is taken
        assert [ah.get(k) for k in (1, 2, 3, 4, 5)] == [2, 4, 6, 8, 10]
        assert ah.get(99) == 198 and ah.get(2) == 4  # not foreseen / asked for again: computed on demand
        ah.close()
        assert all(name.startswith("imcom-psf") for _, name in seen)  # every call into the block's objects is made on a worker thread
        assert [k for k, _ in seen][:5] == [1, 2, 3, 4, 5] or threads > 1


def oracle_order(*a):
    from oracle import oracle as orc

    return orc.stamp_loop_order(*a)
