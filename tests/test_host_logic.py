"""Host-side logic that needs no GPU: stamp neighbourhoods and pivots (coadd.py:853, 918-919), the visiting order of
the pixel partition (coadd.py:329-336), the table sets a stamp needs, and the oracle's own selection / block helpers."""

import numpy as np


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
