"""End to end: one small output block through the device-resident chain (InStamp pool -> selection -> A, B ->
Cholesky kernel -> coaddition -> block maps -> edge recovery) against the oracle's restatement of the reference's
stamp loop (coadd.py:886-977, 1002-1122, 1294-1363, 1939-2001, 2163-2181)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _instamps(cfg, n1P, n_expo, rng):
    """InStamps of the (n1P+2)^2 cells (coadd.py:207, 329-358): per exposure a rotated lattice of native pixels binned
    by cell, exposure-major inside a cell, random data."""
    from pyimcom_amd.synth import NATIVE_ARCSEC

    nst, n2, p = n1P + 2, cfg.n2, NATIVE_ARCSEC / cfg.dtheta_as
    lo, hi = -n2 - 0.5, (n1P + 1) * n2 - 0.5
    cells = [[[] for _ in range(nst)] for _ in range(nst)]
    for e in range(n_expo):
        th = np.deg2rad(11.0 * e + 3.0)
        g = np.arange(-nst * n2, 2 * nst * n2) * p
        xx, yy = np.meshgrid(g + rng.uniform(0, p), g + rng.uniform(0, p))
        x = (np.cos(th) * xx - np.sin(th) * yy).ravel()
        y = (np.sin(th) * xx + np.cos(th) * yy).ravel()
        ok = (x > lo) & (x < hi) & (y > lo) & (y < hi) & (rng.uniform(size=x.size) > 0.01)
        x, y = x[ok], y[ok]
        ci, cj = ((x - lo) // n2).astype(int), ((y - lo) // n2).astype(int)
        for j in range(nst):
            for i in range(nst):
                m = (ci == i) & (cj == j)
                cells[j][i].append((x[m], y[m]))
    out = []
    for j in range(nst):
        for i in range(nst):
            parts = cells[j][i]
            cum = np.concatenate([[0], np.cumsum([len(q[0]) for q in parts])])
            xs, ys = np.hstack([q[0] for q in parts]), np.hstack([q[1] for q in parts])
            out.append((xs, ys, rng.standard_normal((cfg.n_inframe, xs.size)).astype(np.float32), cum))
    return out


def test_block_end_to_end_vs_oracle():
    import torch

    from oracle import oracle as orc
    from pyimcom_amd import smoke, synth
    from pyimcom_amd.blockrun import coadd_block, stamp_neighbours
    from pyimcom_amd.select import InStampPool
    from pyimcom_amd.stamps import PSFGroupTables

    cfg = synth.CONFIGS["tiny"]
    n1P, n_expo = 3, 3
    nst = n1P + 2
    rng = np.random.default_rng(21)
    inst = _instamps(cfg, n1P, n_expo, rng)
    psfs, target = synth.make_psfs(cfg, n_expo)
    tabs = PSFGroupTables(psfs, target, cfg.nfft)
    pool = InStampPool(inst, cfg.n_inframe)
    maps = coadd_block(cfg, pool, tabs, n1P, n_expo, batch=4)  # 9 stamps in three calls
    torch.cuda.synchronize()

    g, _, _ = smoke.oracle_tables(cfg, psfs, target)
    t_gpu = tabs.tables.cpu().numpy()
    pair_tab, pair_pen, io_tab = tabs.pair_maps(cfg.flat_penalty)
    ns = maps.nside
    ref = {k: np.zeros((1, ns, ns), np.float32) for k in ("UC", "Sigma", "kappa", "Tsum", "Neff")}
    ref_out = np.zeros((1, cfg.n_inframe, ns, ns), np.float32)
    nmax = 0
    for j in range(1, n1P + 1):
        for i in range(1, n1P + 1):
            ids, pvx, pvy = stamp_neighbours(j, i, cfg.n2, nst)
            piv = [(None if np.isnan(a) else a, None if np.isnan(b) else b) for a, b in zip(pvx, pvy)]
            x, y, indata, expo, cum = orc.process_input_stamps([inst[k] if k >= 0 else None for k in ids], piv, cfg.rho)
            st = synth.Stamp(x=x, y=y, expo=expo.astype(np.int32), seg=None, indata=indata, out_x0=(i - 1) * cfg.n2 - cfg.fade,
                             out_y0=(j - 1) * cfg.n2 - cfg.fade, n_expo=n_expo, inpix_cumsum=cum)
            nmax = max(nmax, st.n)
            r = smoke.oracle_stamp(cfg, g, t_gpu, tabs.C, st, pair_tab, pair_pen, io_tab)
            orc.block_accumulate(ref_out, r["outimage"][None], j, i, cfg.n2, cfg.fade)
            for name, key in (("UC", "UC"), ("Sigma", "Sigma"), ("kappa", "kappa"), ("Tsum", "Tsum_inpix"), ("Neff", "Neff")):
                orc.block_accumulate(ref[name], np.asarray(r[key], dtype=np.float32)[None], j, i, cfg.n2, cfg.fade)
    orc.trapezoid_recover(ref_out, cfg.fade)
    for name in ref:
        orc.trapezoid_recover(ref[name], cfg.fade)
    assert nmax > 60
    got = maps.out_map.cpu().numpy()
    assert np.abs(got - ref_out[0]).max() <= 5e-5 * np.abs(ref_out[0]).max()
    for name in ref:
        a, b = maps.maps[name].cpu().numpy(), ref[name]
        assert np.allclose(a, b, rtol=2e-5, atol=1e-6 * np.abs(b).max()), (name, np.abs(a - b).max(), np.abs(b).max())
