"""GPU parity at the reference's own benchmark shape (configs/paper4_configs/H158_Chol_benchmark.json -> synth.CONFIGS["paper4"]):
32 x 32-output stamps with FADE 3 (m = 38^2 = 1444), INPAD 1.24" = 31.7 output pixels against n2 = 32 -- one pixel inside the guard of
coadd.py:1915 --, six exposures, six input layers, N ~ 6.2k input pixels per stamp (N / m = 4.3: the factorisation is 42 % of the matrix
flops), kappa / C = 6e-4.  With PSFs of 48 native pixels the overlap tables end at separations of 24 native pixels while the pixels of a
stamp lie up to 34 apart: 11 % of A is cut to zero (psfutil.py:1691-1702 leaves off-table samples untouched), A + kappa I is not
positive definite and EVERY stamp goes through _cholesky_wrapper's repair (lakernel.py:262-279).  The one-stamp A / B / T / maps /
six-layer image test is tests/test_gpu_fullsize.py::test_baseline_config_vs_oracle[paper4-1]."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _corner(n1P, seed=17):
    from pyimcom_amd import synth
    from pyimcom_amd.select import InStampPool

    cfg = synth.CONFIGS["paper4"]
    inst = synth.make_instamps(cfg, n1P, cfg.n_expo, np.random.default_rng(seed))
    return cfg, inst, InStampPool(inst, cfg.n_inframe)


def test_paper4_selection_vs_oracle():
    """ST-1 at rho = 31.744 of n2 = 32 (coadd.py:716-749, 886-977, 923): the edge strips are nearly whole InStamps, the corner quarter
    discs three quarters of one.  Every stamp of a 4 x 4 block -- corners and edges of the block included, where neighbours are
    missing -- bit for bit against the oracle: pixel order, positions, the six layers, exposure indices, segment boundaries."""
    from oracle import oracle as orc
    from pyimcom_amd.blockrun import _neighbours_of
    from pyimcom_amd.select import select_pixels

    n1P = 4
    cfg, inst, pool = _corner(n1P)
    assert 31.7 < cfg.rho < cfg.n2
    nst = n1P + 2
    todo = [(j, i) for j in range(1, n1P + 1) for i in range(1, n1P + 1)]
    nb = _neighbours_of(todo, cfg.n2, nst)
    ids, pvx, pvy = (np.stack([t[q] for t in nb]) for q in range(3))
    cap = max(int(sum(inst[k][0].size for k in row if k >= 0)) for row in ids)
    ld = (cap + 127) // 128 * 128
    x, y, indata, expo, cumsum = select_pixels(pool, ids, pvx, pvy, cfg.rho, ld)
    x, y, indata, expo = x.cpu().numpy(), y.cpu().numpy(), indata.cpu().numpy(), expo.cpu().numpy()
    for s, (j, i) in enumerate(todo):
        piv = [(None if np.isnan(a) else a, None if np.isnan(b) else b) for a, b in zip(pvx[s], pvy[s])]
        rx, ry, rd, re, rc = orc.process_input_stamps([inst[k] if k >= 0 else None for k in ids[s]], piv, cfg.rho)
        n = rx.size
        assert 6000 < n < 6500, n  # every stamp of this block has its nine neighbours (the InStamp array has a guard ring, coadd.py:207)
        assert np.array_equal(cumsum[s], rc), (s, cumsum[s], rc)
        assert np.array_equal(x[s, :n], rx) and np.array_equal(y[s, :n], ry)
        assert np.array_equal(indata[s, :, :n], rd) and np.array_equal(expo[s, :n], re)
        assert not x[s, n:].any() and not indata[s, :, n:].any()
        # an edge strip holds rho / n2 = 99.2 % of its InStamp, a corner pi/4 (rho / n2)^2 = 77 %
        cell = lambda k: inst[ids[s][k]][0].size  # noqa: E731
        seg = np.diff(rc)
        for k in (1, 3, 5, 7):
            assert 0.975 * cell(k) <= seg[k] <= cell(k)
        for k in (0, 2, 6, 8):
            assert 0.72 * cell(k) <= seg[k] <= 0.82 * cell(k)
        assert seg[4] == cell(4)


def test_paper4_block_corner_fade3_six_layers_vs_oracle():
    """A 4 x 4-stamp corner of the production block through ``coadd_block`` (selection, A, B, Cholesky with the repair on every stamp,
    coaddition of six layers, the fade-3 tapers of the maps and of T, overlapping stamps added in the reference's order, boundary
    recovery), planned by the block planner at this stamp size.  The oracle restates the reference's loop for the 2 x 2 stamps at the
    block's corner (four stamps of N ~ 6.2k with an eigendecomposition each: about a minute of host time); the block pixels that only
    those four reach -- the first 64 rows and columns, tapered edges and the recovered boundary included -- are compared with it, the
    rest of the corner through properties."""
    import torch

    from oracle import oracle as orc
    from pyimcom_amd import synth
    from pyimcom_amd.blockrun import coadd_block, plan_block, stamp_neighbours
    from pyimcom_amd.stamps import PSFGroupTables
    from tests import parity

    n1P = 4
    cfg, inst, pool = _corner(n1P)
    E, nst = cfg.n_expo, n1P + 2
    psfs, target = synth.make_psfs(cfg, E)
    tabs = PSFGroupTables(psfs, target, cfg.nfft)
    chunks = plan_block(cfg, pool, tabs, n1P)
    assert sorted(t for c in chunks for t in c) == [(j, i) for j in range(1, n1P + 1) for i in range(1, n1P + 1)]
    maps = coadd_block(cfg, pool, tabs, n1P, E, chunks=chunks, pad_sides="all")
    torch.cuda.synchronize()
    assert maps.info_nonzero == n1P * n1P, "every stamp of this shape needs the repair"
    got = maps.out_map.cpu().numpy()
    assert got.shape == (1, cfg.n_inframe, maps.nside, maps.nside) and np.isfinite(got).all()

    g, _, _ = parity.oracle_tables(cfg, psfs, target)
    t_gpu = tabs.tables.cpu().numpy()
    pair_tab, pair_pen, _ = tabs.pair_maps(cfg.flat_penalty)
    ns = maps.nside
    names = (("UC", "UC"), ("Sigma", "Sigma"), ("kappa", "kappa"), ("Tsum", "Tsum_inpix"), ("Neff", "Neff"))
    ref = {k: np.zeros((1, ns, ns), np.float32) for k, _ in names}
    ref_out = np.zeros((1, cfg.n_inframe, ns, ns), np.float32)
    for j, i in orc.stamp_loop_order(1, 2, 1, 2):
        ids, pvx, pvy = stamp_neighbours(j, i, cfg.n2, nst)
        piv = [(None if np.isnan(a) else a, None if np.isnan(b) else b) for a, b in zip(pvx, pvy)]
        x, y, indata, expo, cum = orc.process_input_stamps([inst[k] if k >= 0 else None for k in ids], piv, cfg.rho)
        st = synth.Stamp(x=x, y=y, expo=expo.astype(np.int32), seg=None, indata=indata, out_x0=(i - 1) * cfg.n2 - cfg.fade,
                         out_y0=(j - 1) * cfg.n2 - cfg.fade, n_expo=E, inpix_cumsum=cum)
        r = parity.oracle_stamp(cfg, g, t_gpu, float(tabs.Cs[0]), st, pair_tab, pair_pen, tabs.io_map(0))
        orc.block_accumulate(ref_out, r["outimage"][None], j, i, cfg.n2, cfg.fade)
        for name, key in names:
            orc.block_accumulate(ref[name], np.asarray(r[key], dtype=np.float32)[None], j, i, cfg.n2, cfg.fade)
    orc.trapezoid_recover(ref_out, cfg.fade)
    for name in ref:
        orc.trapezoid_recover(ref[name], cfg.fade)
    lim = 2 * cfg.n2  # stamp 3 starts at block pixel 2 n2 (its taper's first pixel)
    a, b = got[..., :lim, :lim], ref_out[..., :lim, :lim]
    assert np.abs(b).max() > 0
    # cond ~ 1e5 after the repair: T carries cond x eps ~ 1e-11; the image tolerance of tests/parity.py, summed over <= 4 stamps
    for f in range(cfg.n_inframe):
        assert np.abs(a[0, f] - b[0, f]).max() <= 5e-5 * np.abs(b[0, f]).max(), f
    for name in ref:
        x_, y_ = maps.maps[name].cpu().numpy()[..., :lim, :lim], ref[name][..., :lim, :lim]
        assert np.allclose(x_, y_, rtol=3e-5, atol=1e-6 * np.abs(y_).max()), (name, np.abs(x_ - y_).max(), np.abs(y_).max())
    # the rest of the corner: every layer finite, the white-noise layers of comparable power over the block, kappa = kappaC C + the
    # repair everywhere after the recovery (single kappa node: the map is flat per stamp, its taper sums to one across the overlaps)
    kap = maps.maps["kappa"].cpu().numpy()[0]
    assert np.isfinite(kap).all() and kap.min() > 0
    inner = kap[cfg.fade : ns - cfg.fade, cfg.fade : ns - cfg.fade]
    assert inner.max() / inner.min() < 1.001  # one PSF group: the same kappa C in every stamp, the tapers add up to one
    p_in = np.square(got[0, 1:, :lim, :lim]).mean()
    p_all = np.square(got[0, 1:]).mean()
    assert 0.5 < p_in / p_all < 2.0


def test_paper4_kernel_class_seam_vs_oracle():
    """The drop-in LA kernel class (OutStamp.LAKERNEL["Cholesky"], coadd.py:839-844, 1091-1093) on ONE production-shape stamp handed over as
    host arrays: N = 6.2k, A + kappa I indefinite -- `_cholesky_wrapper`'s repair (lakernel.py:262-279) through the library's small-batch
    path (split-K launches, the smallest eigenvalue by the subspace iteration with one wanted stamp) -- against the oracle's CholKernel
    (scipy cholesky -> LinAlgError -> numpy eigh -> shifted cholesky) on the same A, -B/2, C."""
    import torch

    from oracle import oracle as orc
    from pyimcom_amd import synth
    from pyimcom_amd.lakernel import HipCholKernel
    from pyimcom_amd.stamps import PSFGroupTables, StampBatch
    from tests.golden.make_golden import make_outst

    cfg = synth.CONFIGS["paper4"]
    st = synth.make_stamp(cfg, 3)
    psfs, target = synth.make_psfs(cfg, st.n_expo)
    tabs = PSFGroupTables(psfs, target, cfg.nfft)
    sb = StampBatch(cfg, [st], tabs)
    sb.build()
    torch.cuda.synchronize()
    n, m = st.n, cfg.m
    A = sb.A[0, :n, :n].cpu().numpy().copy()
    mB = np.ascontiguousarray(sb.Bt[0, :n, :m].cpu().numpy().T)[None]
    del sb
    torch.cuda.empty_cache()
    C = np.array([tabs.C])
    kC = np.array(cfg.kappaC)
    o = make_outst(A.copy(), mB.copy(), C, cfg.n2f, kC, cfg.uctarget, cfg.sigmamax)
    K = HipCholKernel(o)
    K()
    assert np.array_equal(o.sysmata, A) and int(K.info[0]) == 1  # A untouched (lakernel.py:277 restores AA, never A); repaired
    To, Uo, So, ko, info_o = orc.chol_kernel(A, mB[0], float(C[0]), kC, cfg.uctarget, cfg.sigmamax)
    assert info_o == 1
    lam_max = float(np.abs(A).sum(axis=1).max())  # (a bound is enough for the tolerance: no second eigendecomposition of a 6.2k matrix)
    kap = kC[0] * C[0]
    cond = (lam_max + kap) / kap  # after the repair the smallest eigenvalue of the factored matrix is kappa + 1e-16
    assert np.abs(o.T[0] - To).max() <= (1e-6 + 50 * cond * 2.2e-16) * np.abs(To).max()
    s2 = (cfg.n2f, cfg.n2f)
    assert np.allclose(o.UC[0], Uo.reshape(s2), rtol=1e-5 + 50 * cond * 2.2e-16, atol=1e-9)
    assert np.allclose(o.Sigma[0], So.reshape(s2), rtol=1e-5 + 50 * cond * 2.2e-16, atol=1e-9)
    assert np.allclose(o.kappa[0], ko.reshape(s2), rtol=1e-5, atol=0)
    # The reference's stamp loop calls the class stamp after stamp: the second call starts its smallest-eigenvalue iteration where the
    # first one's ended (lakernel._repair_memory -> imcom_ctx_set_repair_hint).  Same stamp again: the same answer by the other path.
    from pyimcom_amd.lakernel import _repair_memory

    mem = _repair_memory._recent.get(id(K.ctx))
    assert mem and 1e-6 < mem[-1] < 3e-6
    o2 = make_outst(A.copy(), mB.copy(), C, cfg.n2f, kC, cfg.uctarget, cfg.sigmamax)
    K2 = HipCholKernel(o2)
    K2()
    assert int(K2.info[0]) == 1 and len(mem) >= 2 and abs(mem[-1] - mem[-2]) <= 1e-9 * mem[-1]
    assert np.abs(o2.T[0] - o.T[0]).max() <= 1e-6 * np.abs(o.T[0]).max()
    assert K.ctx.last_repair()[0] == 1 and np.allclose(o2.UC[0], o.UC[0], rtol=1e-5, atol=1e-9)


def test_paper4_block_in_several_passes_expects_the_repair():
    """``coadd_block`` over a 4 x 4 corner in passes of four stamps: the first pass finds every factorisation failing, the following ones
    are told to expect that (StampBatch.solve_begin(expect_repair=True, repair_hint=...)) and go straight to the smallest eigenvalues, all of
    them from the FIRST pass's record (blockrun.RepairRecord: a pass's inputs are a function of the block alone) -- the maps must be
    those of the single-pass block to the rounding of the solves (not bit for bit: launches of different batch sizes deal their tiles
    differently), every stamp repaired.  A block's first pass starts blind whatever ran before it."""
    import torch

    from pyimcom_amd import synth
    from pyimcom_amd.blockrun import coadd_block
    from pyimcom_amd.stamps import PSFGroupTables, StampBatch

    n1P = 4
    cfg, inst, pool = _corner(n1P)
    psfs, target = synth.make_psfs(cfg, cfg.n_expo)
    tabs = PSFGroupTables(psfs, target, cfg.nfft)
    calls = []
    real = StampBatch.solve_begin

    hints = []

    def spy(self, expect_repair=False, repair_hint=None):
        calls.append(bool(expect_repair))
        hints.append(repair_hint)
        return real(self, expect_repair, repair_hint)

    StampBatch.solve_begin = spy
    try:
        one = coadd_block(cfg, pool, tabs, n1P, cfg.n_expo, batch=16)
        assert calls == [False] and hints == [None]
        calls.clear()
        hints.clear()
        state = {}
        four = coadd_block(cfg, pool, tabs, n1P, cfg.n_expo, batch=4, repair_state=state)
        # the record of the block's first pass is handed out (a driver's log) ...
        assert state["share"] == 1.0 and 1e-6 < state["hint"] < 3e-6 and state["hint"] == hints[1]
        n_before = len(calls)
        # ... but not read: the next block's first pass starts blind again, its second pass from ITS first pass's record
        again = coadd_block(cfg, pool, tabs, n1P, cfg.n_expo, batch=8, repair_state=state)
        assert calls[n_before:] == [False, True] and hints[n_before] is None and 1e-6 < hints[n_before + 1] < 3e-6
        assert float((again.out_map - four.out_map).abs().max()) <= 5e-6 * float(four.out_map.abs().max())
        del calls[n_before:], hints[n_before:]
    finally:
        StampBatch.solve_begin = real
    torch.cuda.synchronize()
    assert calls == [False, True, True, True] and one.info_nonzero == four.info_nonzero == 16
    # ... and where the smallest eigenvalues lie (max |w[0]| of the block's first pass: the iteration starts at that shift, one factorisation instead of two)
    assert hints[0] is None and all(h is not None and 1e-6 < h < 3e-6 for h in hints[1:]) and len(set(hints[1:])) == 1, hints
    a, b = one.out_map, four.out_map
    assert float((a - b).abs().max()) <= 5e-6 * float(a.abs().max())
    for k in ("UC", "Sigma", "kappa", "Tsum", "Neff"):
        x, y = one.maps[k], four.maps[k]
        assert torch.allclose(x, y, rtol=2e-5, atol=1e-7 * float(x.abs().max())), k
