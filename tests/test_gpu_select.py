"""GPU parity of the input-pixel selection (reference coadd.py:886-977, 716-749): bit-exact against the oracle on a
synthetic block of InStamps, including block-edge stamps (missing neighbours), empty InStamps and pixels that
sit exactly on the acceptance circle."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _block(rng, nside, n2, n_expo, n_inframe):
    """InStamps of an nside x nside block: per exposure a jittered lattice binned into stamp cells (coadd.py:207)."""
    inst = [[None] * nside for _ in range(nside)]
    cells = {}
    for e in range(n_expo):
        p = 2.2 + 0.1 * e
        g = np.arange(-2, nside * n2 / p + 2)
        xx, yy = np.meshgrid(g * p + rng.uniform(0, p), g * p + rng.uniform(0, p))
        th = np.deg2rad(7.0 * e)
        x = (np.cos(th) * xx - np.sin(th) * yy).ravel()
        y = (np.sin(th) * xx + np.cos(th) * yy).ravel()
        keep = rng.uniform(size=x.size) > 0.02
        x, y = x[keep], y[keep]
        ci, cj = np.floor((x + 0.5) / n2).astype(int), np.floor((y + 0.5) / n2).astype(int)
        for j in range(nside):
            for i in range(nside):
                m = (ci == i) & (cj == j)
                cells.setdefault((j, i), []).append((x[m], y[m], rng.standard_normal((n_inframe, m.sum())).astype(np.float32)))
    for (j, i), parts in cells.items():
        if (j, i) == (1, 2):  # an InStamp without pixels
            parts = [(np.zeros(0), np.zeros(0), np.zeros((n_inframe, 0), np.float32))] * n_expo
        cum = np.concatenate([[0], np.cumsum([len(p_[0]) for p_ in parts])])
        inst[j][i] = (np.hstack([p_[0] for p_ in parts]), np.hstack([p_[1] for p_ in parts]), np.hstack([p_[2] for p_ in parts]), cum)
    return inst


def test_select_pixels_bit_exact():
    from oracle import oracle as orc
    from pyimcom_amd.select import InStampPool, select_pixels

    rng = np.random.default_rng(3)
    nside, n2, n_expo, n_inframe, radius = 4, 8, 3, 2, 3.7
    inst = _block(rng, nside, n2, n_expo, n_inframe)
    # one pixel ON the circle (to rounding) around the lower-left pivot of stamp (1,1): the '<' decision must be
    # the oracle's, i.e. the squares and the sum must round as numpy's do
    x0, y0, d0, c0 = inst[0][0]
    x0[0], y0[0] = (n2 - 0.5) - 3.0 * radius / 5.0, (n2 - 0.5) - 4.0 * radius / 5.0
    flat = [inst[j][i] for j in range(nside) for i in range(nside)]
    pool = InStampPool(flat, n_inframe)
    stamps = [(j, i) for j in range(nside) for i in range(nside)]
    inst_id = np.full((len(stamps), 9), -1, np.int32)
    pvx, pvy = np.full((len(stamps), 9), np.nan), np.full((len(stamps), 9), np.nan)
    refs = []
    for s, (j, i) in enumerate(stamps):
        left, right, bottom, top = i * n2, (i + 1) * n2 - 1, j * n2, (j + 1) * n2 - 1
        nb, piv = [], []
        for idx, (dj, di) in enumerate((dj, di) for dj in (-1, 0, 1) for di in (-1, 0, 1)):
            jj, ii = j + dj, i + di
            xp = [left - 0.5, None, right + 0.5][di + 1]
            yp = [bottom - 0.5, None, top + 0.5][dj + 1]
            piv.append((xp, yp))
            if 0 <= jj < nside and 0 <= ii < nside:
                inst_id[s, idx] = jj * nside + ii
                nb.append(inst[jj][ii])
            else:
                nb.append(None)
            if xp is not None:
                pvx[s, idx] = xp
            if yp is not None:
                pvy[s, idx] = yp
        refs.append(orc.process_input_stamps(nb, piv, radius))
    ldn = max(r[0].size for r in refs) + 5
    x, y, indata, expo, cumsum = select_pixels(pool, inst_id, pvx, pvy, radius, ldn)
    x, y, indata, expo = x.cpu().numpy(), y.cpu().numpy(), indata.cpu().numpy(), expo.cpu().numpy()
    for s, (rx, ry, rd, re, rc) in enumerate(refs):
        n = rx.size
        assert np.array_equal(cumsum[s], rc), (s, cumsum[s], rc)
        assert np.array_equal(x[s, :n], rx) and np.array_equal(y[s, :n], ry)
        assert np.array_equal(indata[s, :, :n], rd) and np.array_equal(expo[s, :n], re)
        assert not x[s, n:].any() and not indata[s, :, n:].any()
    assert refs[5][0].size > 0  # stamp (1,1), whose corner circle passes (to rounding) through the doctored pixel

    with pytest.raises(Exception, match="more than ldn"):
        select_pixels(pool, inst_id, pvx, pvy, radius, 8)


def test_partition_pixels_bit_exact():
    """InImage.partition_pixels binning (coadd.py:329-358): stable partition by stamp, bit-exact vs the oracle loop,
    incl. pixels exactly on stamp boundaries, masked pixels, unused stamps and pixels outside the block."""
    from oracle import oracle as orc
    from pyimcom_amd.select import partition_pixels, visiting_order

    rng = np.random.default_rng(8)
    n1P, n2, npixmax = 4, 6, 260
    nst = n1P + 2
    # a 60 x 60 input image on a sparse grid of 5 x 5 cells, affine map to output pixels (pitch 0.41, rotated)
    sp_arr = np.linspace(0, 60, 6, dtype=np.uint16)
    rel = rng.uniform(size=(5, 5)) > 0.2
    in_y, in_x = visiting_order(rel, sp_arr)
    th = 0.3
    ox = -9.0 + 0.41 * (np.cos(th) * in_x - np.sin(th) * in_y) + 6.0
    oy = -9.0 + 0.41 * (np.sin(th) * in_x + np.cos(th) * in_y)
    ox[:5] = [-6.5, -0.5, 5.5, 11.5, 29.5]   # exactly on stamp boundaries / the lower limit (strict inequality)
    oy[:5] = [3.0, -0.5, 5.5, 3.0, 3.0]
    mask = rng.uniform(size=in_x.size) > 0.1
    use = rng.uniform(size=(nst, nst)) > 0.15
    ref = orc.partition_pixels(ox, oy, in_x, in_y, mask, use, n2, n1P, npixmax)
    got = partition_pixels(ox, oy, in_x, in_y, mask, use, n2, n1P, npixmax)
    assert ref[4].sum() > 500 and ref[4].max() <= npixmax
    for g, r in zip(got, ref):
        assert np.array_equal(g.cpu().numpy(), r)
    with pytest.raises(Exception, match="more than npixmax"):
        partition_pixels(ox, oy, in_x, in_y, mask, use, n2, n1P, 5)


def test_selection_and_coaddition_vs_reference_golden(golden):
    """Device selection (imcom_select_pixels) and coaddition epilogue (imcom_coadd_epilogue) on the inputs of
    tests/golden/coadd_stamp.npz, against what the reference's own _process_input_stamps / _perform_coaddition code
    returned for them.  Selection and the taper of T: bit for bit.  Sums: the reference adds every (InStamp, exposure)
    segment of T in float32 before accumulating in float64 (coadd.py:1329-1337), the device accumulates in float64
    throughout -- a few float32 ulps apart."""
    import ctypes as C

    import torch

    from pyimcom_amd._lib import check, default_context, lib
    from pyimcom_amd.select import InStampPool, select_pixels

    g = golden("coadd_stamp")
    order = [tuple(int(v) for v in ji) for ji in g["sel_order"]]
    flat = [(g[f"in{j}{i}_x"], g[f"in{j}{i}_y"], g[f"in{j}{i}_data"], g[f"in{j}{i}_cum"].astype(np.int64)) for j, i in order]
    n2, fade, n_expo, n_inframe = (int(v) for v in g["co_pars"])
    pool = InStampPool(flat, n_inframe)
    bottom, top, left, right = (int(v) for v in g["sel_box"])
    pvx, pvy = np.full((1, 9), np.nan), np.full((1, 9), np.nan)
    for k, (j, i) in enumerate(order):
        xp, yp = [left - 0.5, None, right + 0.5][i - 2 + 1], [bottom - 0.5, None, top + 0.5][j - 2 + 1]
        if xp is not None:
            pvx[0, k] = xp
        if yp is not None:
            pvy[0, k] = yp
    N = int(g["sel_cumsum"][-1])
    ldn = 256
    x, y, indata, expo, cumsum = select_pixels(pool, np.arange(9, dtype=np.int32)[None], pvx, pvy, float(g["sel_rpix"]), ldn)
    assert np.array_equal(cumsum[0], g["sel_cumsum"].astype(cumsum.dtype))
    assert np.array_equal(x[0, :N].cpu().numpy(), g["sel_x"]) and np.array_equal(y[0, :N].cpu().numpy(), g["sel_y"])
    assert np.array_equal(indata[0, :, :N].cpu().numpy(), g["sel_data"])

    n2f = n2 + 2 * fade
    m, ldm = n2f * n2f, 128
    ctx = default_context()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    dev = x.device
    dp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    for o in range(g["co_T_in"].shape[0]):
        Tt = torch.zeros((1, ldn, ldm), dtype=torch.float32, device=dev)
        Tt[0, :N, :m] = torch.as_tensor(g["co_T_in"][o].T.copy(), device=dev)
        outimage = torch.empty((1, n_inframe, m), dtype=torch.float32, device=dev)
        Ts = torch.empty((1, n_expo), dtype=torch.float64, device=dev)
        Tin = torch.empty((1, m), dtype=torch.float64, device=dev)
        Neff = torch.empty((1, m), dtype=torch.float64, device=dev)
        nn = np.array([N], np.int32)
        check(lib.imcom_coadd_epilogue(ctx.handle, 1, nn.ctypes.data_as(C.c_void_p), ldn, m, ldm, n2f, fade, n2, dp(Tt), dp(indata), n_inframe,
                                       dp(expo), n_expo, dp(outimage), dp(Ts), dp(Tin), dp(Neff)))
        torch.cuda.synchronize()
        assert np.array_equal(Tt[0, :N, :m].cpu().numpy().T, g["co_T_out"][o])  # tapered T, float32
        ref_img = g["co_outimage"][o].reshape(n_inframe, m)
        assert np.abs(outimage[0].cpu().numpy() - ref_img).max() <= 1e-6 * np.abs(ref_img).max()
        assert np.allclose(Ts[0].cpu().numpy(), g["co_Tsum_stamp"][o], rtol=1e-6, atol=0)
        assert np.allclose(Tin[0].cpu().numpy(), g["co_Tsum_inpix"][o].ravel(), rtol=0, atol=2e-7 * np.abs(g["co_Tsum_inpix"][o]).max())
        assert np.allclose(Neff[0].cpu().numpy(), g["co_Neff"][o].ravel(), rtol=2e-6, atol=0)


def test_partition_vs_reference_golden(golden):
    """imcom_partition_pixels against the arrays the reference's own binning statement filled (partition.npz)."""
    from pyimcom_amd.select import partition_pixels
    from tests.test_host_logic import _partition_inputs

    g = golden("partition")
    in_y, in_x, ox, oy, mask = _partition_inputs(g)
    got = partition_pixels(ox, oy, in_x, in_y, mask, g["use_instamps"], int(g["n2"]), int(g["n1P"]), int(g["npixmax"]))
    for a, name in zip(got, ("y_idx", "x_idx", "y_val", "x_val", "pix_count")):
        assert np.array_equal(a.cpu().numpy(), g[name]), name


@pytest.mark.parametrize("n1P,npix", [(10, 300_001), (62, 1_000_003)])
def test_partition_large_stable(n1P, npix):
    """The binning at image scale: many workgroups of the radix sort, two passes (144 stamps) and three (4096), random
    positions with a quarter of the pixels dropped (outside, masked, unused stamps); every stamp's pixels must be exactly the
    stamp's pixels in visiting order (numpy's stable argsort is the reference)."""
    import torch

    from pyimcom_amd.select import partition_pixels

    rng = np.random.default_rng(n1P)
    n2, nst = 8, n1P + 2
    lo, hi = -n2 - 0.5, n1P * n2 + n2 - 0.5
    ox, oy = rng.uniform(lo - 3, hi + 3, npix), rng.uniform(lo - 3, hi + 3, npix)
    in_x, in_y = rng.integers(0, 4088, npix).astype(np.uint16), rng.integers(0, 4088, npix).astype(np.uint16)
    mask = rng.uniform(size=npix) > 0.1
    use = rng.uniform(size=(nst, nst)) > 0.05
    ist, jst = np.floor((ox - lo) / n2).astype(int), np.floor((oy - lo) / n2).astype(int)
    ok = (lo < ox) & (ox < hi) & (lo < oy) & (oy < hi) & mask
    ok &= use[np.clip(jst, 0, nst - 1), np.clip(ist, 0, nst - 1)]
    key = np.where(ok, jst * nst + ist, nst * nst)
    order = np.argsort(key, kind="stable")
    counts = np.bincount(key, minlength=nst * nst + 1)[: nst * nst]
    npixmax = int(counts.max())
    y_idx, x_idx, y_val, x_val, count = partition_pixels(ox, oy, in_x, in_y, mask, use, n2, n1P, npixmax)
    torch.cuda.synchronize()
    assert np.array_equal(count.cpu().numpy().astype(np.int64).ravel(), counts)
    got_x, got_ix = x_val.cpu().numpy().reshape(nst * nst, npixmax), x_idx.cpu().numpy().reshape(nst * nst, npixmax)
    got_y = y_val.cpu().numpy().reshape(nst * nst, npixmax)
    start = np.concatenate([[0], np.cumsum(counts)])
    for k in rng.choice(nst * nst, 60, replace=False):
        sel = order[start[k] : start[k + 1]]
        assert np.array_equal(got_x[k, : counts[k]], ox[sel]) and np.array_equal(got_y[k, : counts[k]], oy[sel]) and np.array_equal(got_ix[k, : counts[k]], in_x[sel])
    with pytest.raises(Exception):
        partition_pixels(ox, oy, in_x, in_y, mask, use, n2, n1P, npixmax - 1)
