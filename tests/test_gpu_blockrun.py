"""End to end: one small output block through the device-resident chain (InStamp pool -> selection -> A, B ->
Cholesky kernel -> coaddition -> block maps -> edge recovery) against the oracle's restatement of the reference's
stamp loop (coadd.py:886-977, 1002-1122, 1294-1363, 1939-2001, 2163-2181)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _instamps(cfg, n1P, n_expo, rng):
    from pyimcom_amd.synth import make_instamps

    return make_instamps(cfg, n1P, n_expo, rng)


@pytest.mark.parametrize("n_out", [1, 2])
def test_block_end_to_end_vs_oracle(n_out):
    import dataclasses

    import torch

    from oracle import oracle as orc
    from pyimcom_amd import synth
    from tests import parity as smoke
    from pyimcom_amd.blockrun import coadd_block, stamp_neighbours
    from pyimcom_amd.select import InStampPool
    from pyimcom_amd.stamps import PSFGroupTables

    cfg = dataclasses.replace(synth.CONFIGS["tiny"], n_out=n_out)
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
    pair_tab, pair_pen, _ = tabs.pair_maps(cfg.flat_penalty)
    ns = maps.nside
    ref = {k: np.zeros((n_out, ns, ns), np.float32) for k in ("UC", "Sigma", "kappa", "Tsum", "Neff")}
    ref_out = np.zeros((n_out, cfg.n_inframe, ns, ns), np.float32)
    nmax = 0
    for j in range(1, n1P + 1):
        for i in range(1, n1P + 1):
            ids, pvx, pvy = stamp_neighbours(j, i, cfg.n2, nst)
            piv = [(None if np.isnan(a) else a, None if np.isnan(b) else b) for a, b in zip(pvx, pvy)]
            x, y, indata, expo, cum = orc.process_input_stamps([inst[k] if k >= 0 else None for k in ids], piv, cfg.rho)
            st = synth.Stamp(x=x, y=y, expo=expo.astype(np.int32), seg=None, indata=indata, out_x0=(i - 1) * cfg.n2 - cfg.fade,
                             out_y0=(j - 1) * cfg.n2 - cfg.fade, n_expo=n_expo, inpix_cumsum=cum)
            nmax = max(nmax, st.n)
            rs = [smoke.oracle_stamp(cfg, g, t_gpu, float(tabs.Cs[o]), st, pair_tab, pair_pen, tabs.io_map(o)) for o in range(n_out)]
            orc.block_accumulate(ref_out, np.stack([r["outimage"] for r in rs]), j, i, cfg.n2, cfg.fade)
            for name, key in (("UC", "UC"), ("Sigma", "Sigma"), ("kappa", "kappa"), ("Tsum", "Tsum_inpix"), ("Neff", "Neff")):
                orc.block_accumulate(ref[name], np.stack([np.asarray(r[key], dtype=np.float32) for r in rs]), j, i, cfg.n2, cfg.fade)
    orc.trapezoid_recover(ref_out, cfg.fade)
    for name in ref:
        orc.trapezoid_recover(ref[name], cfg.fade)
    assert nmax > 60
    got = maps.out_map.cpu().numpy()
    assert got.shape == ref_out.shape
    assert np.abs(got - ref_out).max() <= 5e-5 * np.abs(ref_out).max()
    if n_out > 1:
        assert np.abs(got[0] - got[1]).max() > 1e-3 * np.abs(got[0]).max()
    for name in ref:
        a, b = maps.maps[name].cpu().numpy(), ref[name]
        assert np.allclose(a, b, rtol=2e-5, atol=1e-6 * np.abs(b).max()), (name, np.abs(a - b).max(), np.abs(b).max())


@pytest.mark.parametrize("n_out", [1, 2])
def test_block_with_psf_groups_vs_oracle(n_out):
    """PSFs that differ between the 2x2 groups of InStamps: per-stamp pair maps, cross-group tables, table arena.
    The oracle assembles A and B the reference's way -- sub-block by sub-block from PSFOvl(group, group') through
    _call_ii_self / _call_ii_cross / _call_io_cross (functions pinned by the reference's goldens), an independent
    route from the device's per-pixel pair codes."""
    import dataclasses

    import torch

    from oracle import oracle as orc
    from pyimcom_amd import synth
    from tests import parity as smoke
    from pyimcom_amd.blockrun import coadd_block, stamp_neighbours
    from pyimcom_amd.select import InStampPool
    from pyimcom_amd.stamps import BlockTables

    cfg = dataclasses.replace(synth.CONFIGS["tiny"], n_out=n_out)
    n1P, n_expo = 2, 3
    nst = n1P + 2
    rng = np.random.default_rng(33)
    inst = _instamps(cfg, n1P, n_expo, rng)
    base, target = synth.make_psfs(cfg, n_expo)
    ns = base.shape[-1]
    lin = np.arange(ns) - ns // 2
    group_psfs = {}
    for gj in range(nst // 2):
        for gi in range(nst // 2):  # a different smooth modulation per group (kept positive, renormalised)
            mod = 1.0 + 0.08 * (gj + 1) * np.cos(0.21 * lin)[None, :, None] * 0 + 0.06 * (gi + 1) * np.sin(0.17 * lin)[None, None, :]
            mod = mod + 0.05 * (gj + 1) * np.cos(0.13 * lin)[None, :, None]
            p = base * mod
            group_psfs[(gj, gi)] = p / p.sum(axis=(1, 2), keepdims=True)
    # group (0, 1) holds PSFs for exposures 0 and 2 only (the reference drops exposures without pixels in the group,
    # psfutil.py:812-832): its four InStamps lose their exposure-1 pixels
    group_expo = {k: [0, 1, 2] for k in group_psfs}
    group_expo[(0, 1)] = [0, 2]
    group_psfs[(0, 1)] = group_psfs[(0, 1)][[0, 2]]
    for jj in (0, 1):
        for ii in (2, 3):
            xs, ys, dat, cum = inst[jj * nst + ii]
            keep = np.r_[0 : cum[1], cum[2] : cum[3]]
            inst[jj * nst + ii] = (xs[keep], ys[keep], dat[:, keep], np.array([0, cum[1], cum[1], cum[1] + cum[3] - cum[2]]))
    tabs = BlockTables(group_psfs, target, cfg.nfft, group_expo=group_expo, capacity=96)
    pool = InStampPool(inst, cfg.n_inframe)
    maps = coadd_block(cfg, pool, tabs, n1P, n_expo, batch=2)
    torch.cuda.synchronize()

    geo = orc.Geom(cfg.npixpsf, cfg.oversamp, cfg.dtheta_as / 3600.0, cfg.flat_penalty)
    rft_in = {k: orc.pad_and_rfft2(v, geo) for k, v in group_psfs.items()}
    rft_out = orc.pad_and_rfft2(target, geo)
    Cs = np.asarray(orc.overlap_out_C(rft_out, geo), dtype=np.float64)
    nsd = maps.nside
    ref = {k: np.zeros((n_out, nsd, nsd), np.float32) for k in ("UC", "Sigma", "kappa", "Tsum", "Neff")}
    ref_out = np.zeros((n_out, cfg.n_inframe, nsd, nsd), np.float32)
    g1 = np.arange(cfg.n2f, dtype=np.float64)
    for j in range(1, n1P + 1):
        for i in range(1, n1P + 1):
            ids, pvx, pvy = stamp_neighbours(j, i, cfg.n2, nst)
            piv = [(None if np.isnan(a) else a, None if np.isnan(b) else b) for a, b in zip(pvx, pvy)]
            nine = [inst[k] if k >= 0 else None for k in ids]
            sels = [None if t is None else orc.select_pixels(t[0], t[1], pv, cfg.rho) for t, pv in zip(nine, piv)]
            groups = [None if k < 0 else (int(k) // nst >> 1, int(k) % nst >> 1) for k in ids]
            x, y, indata, expo, cum = orc.process_input_stamps(nine, piv, cfg.rho)
            ox, oy = (i - 1) * cfg.n2 - cfg.fade + g1, (j - 1) * cfg.n2 - cfg.fade + g1
            for o in range(n_out):  # the oracle's assembly serves one target at a time
                A, mB = orc.stamp_system_groups(nine, sels, groups, rft_in, rft_out[o : o + 1], geo, ox, oy, group_expo)
                T, UC, Sg, kp, _ = orc.chol_kernel(A, mB, float(Cs[o]), np.array(cfg.kappaC), cfg.uctarget, cfg.sigmamax)
                s2 = (cfg.n2f, cfg.n2f)
                UC, Sg, kp = UC.reshape(s2).copy(), Sg.reshape(s2).copy(), kp.reshape(s2).copy()
                for a in (kp, Sg, UC):
                    orc.trapezoid(a, cfg.fade)
                T3 = T[None].copy()
                outimage, _, Tin, Neff = orc.perform_coaddition(T3, indata, expo, n_expo, cfg.n2f, cfg.n2, cfg.fade)
                orc.block_accumulate(ref_out[o : o + 1], outimage, j, i, cfg.n2, cfg.fade)
                for name, v in (("UC", UC), ("Sigma", Sg), ("kappa", kp), ("Tsum", Tin[0]), ("Neff", Neff[0])):
                    orc.block_accumulate(ref[name][o : o + 1], np.asarray(v, dtype=np.float32)[None], j, i, cfg.n2, cfg.fade)
    orc.trapezoid_recover(ref_out, cfg.fade)
    for name in ref:
        orc.trapezoid_recover(ref[name], cfg.fade)
    assert np.abs(tabs.Cs - Cs).max() <= 1e-12 * Cs.min()
    got = maps.out_map.cpu().numpy()
    assert np.abs(got - ref_out).max() <= 5e-5 * np.abs(ref_out).max()
    for name in ref:
        a, b = maps.maps[name].cpu().numpy(), ref[name]
        assert np.allclose(a, b, rtol=5e-5, atol=2e-6 * np.abs(b).max()), (name, np.abs(a - b).max(), np.abs(b).max())
    # the groups matter: the same block with one PSF group for all stamps is far outside these tolerances
    from pyimcom_amd.stamps import PSFGroupTables

    uni = coadd_block(cfg, pool, PSFGroupTables(group_psfs[(0, 0)], target[:1], cfg.nfft), n1P, n_expo, batch=4)  # 3 PSFs for all
    assert np.abs(uni.out_map[0].cpu().numpy() - ref_out[0]).max() > 1e-3 * np.abs(ref_out[0]).max()


@pytest.mark.parametrize("name", ["stamp_chain", "stamp_chain_mid"])
def test_full_chain_vs_reference_golden(golden, name):
    """The device chain from raw inputs -- PSF images sampled on the device, PSF groups per 2x2 InStamps (one lacking an
    exposure), table sets, selection, A, B, Cholesky, taper, coaddition -- for the output stamp of
    tests/golden/stamp_chain.npz, against what the reference's own code produced end to end (PSFGrp, PSFOvl, SysMatA /
    SysMatB, OutStamp._build_system_matrices, CholKernel, _perform_coaddition)."""
    import torch

    from pyimcom_amd import psfs, synth
    from pyimcom_amd.blockrun import prepare_batch
    from pyimcom_amd.select import InStampPool
    from pyimcom_amd.stamps import BlockTables
    from tests.test_oracle import _chain_inputs

    g = golden(name)
    geo, inst, _, group_expo, (n1P, n2, fade, n_inimage, n_inframe) = _chain_inputs(g)
    ns, nst = geo.nsamp, n1P + 2
    cfg = synth.WorkloadConfig("chain", n2, fade, float(g["dtheta_as"]), n_inimage, float(g["instamp_pad_as"]), "Cholesky",
                               tuple(float(k) for k in g["kappaC"]), npixpsf=int(g["npixpsf"]), oversamp=int(g["oversamp"]),
                               flat_penalty=float(g["flat_penalty"]), n_inframe=n_inframe, uctarget=1e-6, sigmamax=0.5)
    assert abs(cfg.dscale - geo.dscale) <= 1e-15 * geo.dscale
    dev = "cuda:0"
    lin = np.arange(ns) - (ns - 1) / 2.0
    gx, gy = np.meshgrid(lin, lin)
    xy = np.stack([gx.ravel(), gy.ravel()], axis=1) * geo.dscale
    group_psfs = {}
    for (gj, gi), expos in group_expo.items():  # PSFGrp._build_inpsfgrp / _sample_psf on the device
        p0 = np.array([2 * gi * n2 - 0.5, 2 * gj * n2 - 0.5])
        imgs = np.stack([g[f"inpsf{e}"] for e in expos])
        co = []
        for e in expos:
            M, t0 = g[f"inM{e}"], g[f"int0{e}"]
            d = ((xy + p0) @ M.T + t0 - (p0[None, :] @ M.T + t0)) * geo.oversamp
            co.append(np.stack([d[:, 1].reshape(ns, ns), d[:, 0].reshape(ns, ns)]))
        group_psfs[(gj, gi)] = psfs.sample_psf(torch.as_tensor(imgs, device=dev), ns, torch.as_tensor(np.stack(co), device=dev), True, True)
    timg = psfs.get_outpsf("GAUSSIAN", 1.1, 2, ns, geo.oversamp, device=dev)
    target = psfs.sample_psf(timg[None], ns, None, True, True)
    tabs = BlockTables({k: v.cpu().numpy() for k, v in group_psfs.items()}, target.cpu().numpy(), geo.nfft, group_expo=group_expo, capacity=256)
    assert abs(tabs.C - float(g["C"][0])) <= 1e-12 * float(g["C"][0])
    pool = InStampPool([inst[(j, i)] for j in range(nst) for i in range(nst)], n_inframe)
    j_st, i_st = int(g["j_st"]), int(g["i_st"])
    sb = prepare_batch(cfg, pool, tabs, [(j_st, i_st)], n1P, n_inimage)
    res = sb.run()
    torch.cuda.synchronize()
    N, m = g["A"].shape[0], cfg.m
    assert int(sb.n[0]) == N
    # (the batch orders a stamp's pixels by PSF, blockrun.prepare_batch: back to the reference's order of coadd.py:937 for the comparison)
    assert sb.perm is not None and sorted(sb.perm[0, :N].tolist()) == list(range(N))
    A = res.to_reference_order(0, sb.A[0, :N, :N], axes=(0, 1)).cpu().numpy()
    Bt = res.to_reference_order(0, sb.Bt[0, :N, :m], axes=(0,)).cpu().numpy()
    assert np.abs(A - g["A"]).max() <= 1e-11 * np.abs(g["A"]).max()
    assert np.abs(Bt.T - g["mBhalf"][0]).max() <= 1e-11 * np.abs(g["mBhalf"]).max()
    lam = np.linalg.eigvalsh(g["A"])
    kap = float(g["kappaC"][0]) * float(g["C"][0])
    cond = (lam[-1] + kap) / (max(lam[0], 0.0) + kap)
    T = res.T(0).cpu().numpy()
    assert np.abs(T - g["T"][0]).max() <= (1e-6 + 50 * cond * 2.2e-16) * np.abs(g["T"]).max()
    for name in ("UC", "Sigma", "kappa"):
        assert np.allclose(getattr(res, name)[0].cpu().numpy(), g[name][0], rtol=1e-5 + 50 * cond * 2.2e-16, atol=1e-9), name
    ref_img = g["outimage"][0]
    assert np.abs(res.outimage[0].cpu().numpy() - ref_img).max() <= 2e-5 * np.abs(ref_img).max()
    assert np.allclose(res.Tsum_stamp[0, :n_inimage].cpu().numpy(), g["Tsum_stamp"][0], rtol=1e-5)
    assert np.allclose(res.Neff[0].cpu().numpy(), g["Neff"][0], rtol=1e-4)


def test_block_tables_eviction_policy():
    """The table arena says what it does when full: 'evict' frees the least recently used sets the request itself does not
    need, recomputes on demand and counts it; 'raise' refuses; a request that can never fit raises under both."""
    import torch

    from pyimcom_amd import synth
    from pyimcom_amd.stamps import BlockTables

    cfg = synth.CONFIGS["tiny"]
    psfs, target = synth.make_psfs(cfg, 3)
    groups = {(0, 0): psfs, (0, 1): psfs * 1.0, (1, 0): psfs[:2]}
    a, b = BlockTables.keys_for([(0, 0), (0, 1)]), BlockTables.keys_for([(1, 0)])
    same = lambda x, y: x.keys() == y.keys() and all(np.array_equal(x[k], y[k]) for k in x)  # noqa: E731
    for policy in ("evict", "raise"):
        t = BlockTables(groups, target, cfg.nfft, capacity=27, on_full=policy)  # the sets of `a` need 6 + 6 + 3 + 3 + 9 = 27 tables
        first = {k: v.copy() for k, v in t.require(a).items()}
        assert t.used == 27 and t.evictions == 0 and same(t.require(a), first)
        assert sorted(np.concatenate(list(first.values())).tolist()) == list(range(1, 28))  # table 0 is the arena's zero table
        keep = {k: t.tables_of(k).clone() for k in a}
        if policy == "evict":
            got = t.require(b)  # 3 + 2 tables: the least recently used set of `a` (self overlap of (0, 0), 6 tables) makes room
            assert t.evictions == 1 and t.evicted_tables == 6 and t.used == 26 and ("self", (0, 0)) not in t.slots
            assert sorted(np.concatenate(list(got.values())).tolist()) == sorted(first[("self", (0, 0))].tolist())[:5]
            ref = BlockTables(groups, target, cfg.nfft, capacity=64)
            ref.require(b)
            for k in b:  # recomputed tables are the same tables
                assert torch.equal(t.tables_of(k), ref.tables_of(k))
            for k in a[1:]:  # the sets that stayed are untouched
                assert torch.equal(t.tables_of(k), keep[k])
            t.require(a)  # one free table + the five of `b` (not wanted now): room for the six that come back
            assert t.evictions == 2 and t.evicted_tables == 11 and t.used == 27 and not any(k in t.slots for k in b)
            for k in a:
                assert torch.equal(t.tables_of(k), keep[k])
        else:
            with pytest.raises(ValueError, match="arena full"):
                t.require(b)
        with pytest.raises(ValueError, match="cannot hold"):
            BlockTables(groups, target, cfg.nfft, capacity=20, on_full=policy).require(a)


def test_block_tables_spectra_arena_reset():
    """The spectra of the PSF groups live in an arena of their own: when its rows are used up, the groups the current request
    does not need are dropped and come back (re-uploaded / re-sampled, re-transformed) when asked for again -- same tables."""
    import torch

    from pyimcom_amd import synth
    from pyimcom_amd.stamps import BlockTables

    cfg = synth.CONFIGS["tiny"]
    psfs, target = synth.make_psfs(cfg, 3)
    groups = {(0, q): psfs * (1.0 + 0.1 * q) for q in range(4)}
    ref = BlockTables(groups, target, cfg.nfft, capacity=200)
    t = BlockTables(groups, target, cfg.nfft, capacity=200, spec_capacity=1 + 2 * 3)  # the target's row + two groups
    for q, pair in enumerate((((0, 0), (0, 1)), ((0, 2), (0, 3)), ((0, 1), (0, 2)), ((0, 0), (0, 1)))):
        keys = BlockTables.keys_for(pair)
        if q == 3:
            t.drop_all()  # the tables of the first request are still resident: forget them, so that they are rebuilt from re-transformed spectra
        t.require(keys)
        ref.require(keys)
        for k in keys:
            assert torch.equal(t.tables_of(k), ref.tables_of(k)), k
    assert t.spectra_resets == 3 and ref.spectra_resets == 0
    with pytest.raises(ValueError, match="spectra arena"):
        t.require(BlockTables.keys_for([(0, 0), (0, 1), (0, 2)]))


@pytest.mark.gpu
def test_block_tables_group_sources_agree():
    """A group's sampled PSFs may come as a host array, as a device tensor, or from a callable evaluated when the group is
    first needed (with ``group_count``): the three give the same tables, and the callable of a group nobody asks for is
    never run."""
    import torch

    from pyimcom_amd import synth
    from pyimcom_amd.stamps import BlockTables

    cfg = synth.CONFIGS["tiny"]
    psfs, target = synth.make_psfs(cfg, 3)
    host = {(0, 0): psfs, (0, 1): psfs[::-1].copy(), (1, 1): psfs[:2]}
    keys = BlockTables.keys_for([(0, 0), (0, 1)])
    ref = BlockTables(host, target, cfg.nfft, capacity=64)
    want = ref.require(keys)
    same = lambda x, y: x.keys() == y.keys() and all(np.array_equal(x[k], y[k]) for k in x)  # noqa: E731
    dev = {k: torch.as_tensor(v, device="cuda:0") for k, v in host.items()}
    called = []

    def provider(k):
        def f():
            called.append(k)
            return torch.as_tensor(host[k], device="cuda:0")
        return f

    for groups, count in ((dev, None), ({k: provider(k) for k in host}, {k: v.shape[0] for k, v in host.items()})):
        t = BlockTables(groups, target, cfg.nfft, capacity=64, group_count=count)
        got = t.require(keys)
        assert same(got, want) and torch.equal(t.tables[1 : t.used + 1], ref.tables[1 : ref.used + 1])
    assert sorted(called) == [(0, 0), (0, 1)]
    # a bulk provider serves runs of neighbouring groups in one call each (and only the groups somebody asked for)
    runs = []

    def bulk(ks):
        runs.append(list(ks))
        return torch.cat([torch.as_tensor(host[k], device="cuda:0") for k in ks]).contiguous()

    t = BlockTables({k: None for k in host}, target, cfg.nfft, capacity=64, group_count={k: v.shape[0] for k, v in host.items()},
                    bulk_provider=bulk)
    assert same(t.require(keys), want) and torch.equal(t.tables[1 : t.used + 1], ref.tables[1 : ref.used + 1])
    assert runs == [[(0, 0), (0, 1)]]
    t.require(BlockTables.keys_for([(1, 1)]))
    assert runs == [[(0, 0), (0, 1)], [(1, 1)]]


@pytest.mark.gpu
def test_ragged_batches_are_reordered_without_a_trace(monkeypatch):
    """A block whose exposure depth varies: prepare_batch visits the deepest stamps first (XCD balance of the late block rows).
    The maps are placed by coordinates and a stamp's arithmetic does not depend on its position in the batch, so the block comes
    out bit for bit as in coordinate order (IMCOM_BLOCK_KEEP_ORDER=1)."""
    import torch

    from pyimcom_amd import synth
    from pyimcom_amd.blockrun import coadd_block, prepare_batch
    from pyimcom_amd.select import InStampPool
    from pyimcom_amd.stamps import PSFGroupTables

    cfg = synth.CONFIGS["small"]
    n1P, n_expo = 4, 4
    nst = n1P + 2
    inst = _instamps(cfg, n1P, n_expo, np.random.default_rng(77))
    for jj in range(nst):  # the right half of the block loses exposures 2 and 3: half the depth there
        for ii in range(nst // 2, nst):
            xs, ys, dat, cum = inst[jj * nst + ii]
            k = int(cum[2])
            inst[jj * nst + ii] = (xs[:k], ys[:k], dat[:, :k], np.array([0, cum[1], cum[2], cum[2], cum[2]]))
    pool = InStampPool(inst, cfg.n_inframe)
    psfs, target = synth.make_psfs(cfg, n_expo)
    tabs = PSFGroupTables(psfs, target, cfg.nfft)
    chunk = [(j, i) for j in range(1, n1P + 1) for i in range(1, n1P + 1)]
    sb = prepare_batch(cfg, pool, tabs, chunk, n1P, n_expo)
    assert sb.chunk != chunk and sorted(sb.chunk) == chunk and np.all(np.diff(sb.n.astype(np.int64)) <= 0)
    assert (sb.n.max() + 127) // 128 > (sb.n.min() + 127) // 128, "the test block is not ragged in 128-blocks"
    a = coadd_block(cfg, pool, tabs, n1P, n_expo, batch=16)
    monkeypatch.setenv("IMCOM_BLOCK_KEEP_ORDER", "1")
    assert prepare_batch(cfg, pool, tabs, chunk, n1P, n_expo).chunk == chunk
    b = coadd_block(cfg, pool, tabs, n1P, n_expo, batch=16)
    torch.cuda.synchronize()
    assert torch.equal(a.out_map, b.out_map) and torch.equal(a.T_weightmap, b.T_weightmap)
    for k in ("UC", "Sigma", "kappa", "Tsum", "Neff"):
        assert torch.equal(a.maps[k], b.maps[k]), k
    assert float(a.out_map.abs().max()) > 0


def test_a_pass_that_runs_out_of_memory_is_run_in_halves(monkeypatch):
    """The plan of a block is an estimate of what fits; a pass whose solve reports IMCOM_ERR_NOMEM (or a launch HIP refused for lack of
    memory) is run again as two passes of half the stamps, cut on a cell boundary, and the block comes out as if those halves had been
    planned: the same bits as the explicit plan.  Both kinds of solve: the two halves of the Cholesky kernel and the synchronous one."""
    import dataclasses

    import torch

    from pyimcom_amd import synth
    from pyimcom_amd._lib import ImcomError
    from pyimcom_amd.blockrun import coadd_block
    from pyimcom_amd.select import InStampPool
    from pyimcom_amd.stamps import PSFGroupTables, StampBatch

    n1P, n_expo = 4, 3
    for kernel in ("Cholesky", "Eigen"):
        cfg = dataclasses.replace(synth.CONFIGS["tiny"], kernel=kernel)
        inst = _instamps(cfg, n1P, n_expo, np.random.default_rng(5))
        psfs, target = synth.make_psfs(cfg, n_expo)
        tabs = PSFGroupTables(psfs, target, cfg.nfft)
        pool = InStampPool(inst, cfg.n_inframe)
        todo = [(j, i) for j in range(1, n1P + 1) for i in range(1, n1P + 1)]
        plan = [todo[:8], todo[8:]]
        real = StampBatch.solve_end

        def refuse(self):  # passes of more than four stamps do not "fit"
            if self.batch > 4:
                self._deferred = self._unsolved = False
                if self.cfg.kernel == "Cholesky":
                    self.ctx.release_workspace()  # (forget the begin that is outstanding)
                raise ImcomError(-3, "device workspace: out of memory (injected)")
            return real(self)

        with monkeypatch.context() as mp:
            mp.setenv("IMCOM_SOLVE_DEFERRED", "0")  # (the injected failure stands for both; an outstanding begin would have to be ended)
            mp.setattr(StampBatch, "solve_end", refuse)
            got = coadd_block(cfg, pool, tabs, n1P, n_expo, chunks=plan)
            torch.cuda.synchronize()
        assert got.passes_halved == 2
        ref = coadd_block(cfg, pool, tabs, n1P, n_expo, chunks=[todo[:4], todo[4:8], todo[8:12], todo[12:]])
        torch.cuda.synchronize()
        assert ref.passes_halved == 0
        assert torch.equal(got.out_map, ref.out_map) and torch.equal(got.T_weightmap, ref.T_weightmap), kernel
        for k in ("UC", "Sigma", "kappa", "Tsum", "Neff"):
            assert torch.equal(got.maps[k], ref.maps[k]), (kernel, k)


@pytest.mark.parametrize("with_claim", [False, True])
@pytest.mark.parametrize("site", ["solve_begin", "next_batch", "solve_end"])
def test_out_of_memory_at_any_point_of_a_pass_loses_no_pass(monkeypatch, site, with_claim):
    """ADVICE r05 (high / medium): the out-of-memory handler of coadd_block covers solve_begin, the preparation of the NEXT pass and
    solve_end.  Wherever the failure comes from, every pass this process took ends up in the maps -- a pass whose preparation failed is
    prepared again, not skipped (it was already claimed: under the farm nobody else would ever coadd it) -- and the block has the bits
    of a clean run.  Standalone and with a claim callback (the farm's)."""
    import torch

    from pyimcom_amd import blockrun, synth
    from pyimcom_amd._lib import ImcomError
    from pyimcom_amd.blockrun import coadd_block
    from pyimcom_amd.select import InStampPool
    from pyimcom_amd.stamps import PSFGroupTables, StampBatch

    n1P, n_expo = 4, 3
    cfg = synth.CONFIGS["tiny"]
    inst = _instamps(cfg, n1P, n_expo, np.random.default_rng(5))
    psfs, target = synth.make_psfs(cfg, n_expo)
    tabs = PSFGroupTables(psfs, target, cfg.nfft)
    pool = InStampPool(inst, cfg.n_inframe)
    todo = [(j, i) for j in range(1, n1P + 1) for i in range(1, n1P + 1)]
    plan = [todo[:8], todo[8:]]
    asked = []

    def claim(q):
        asked.append(q)
        return True

    fired = []
    real_begin, real_end, real_prepare = StampBatch.solve_begin, StampBatch.solve_end, blockrun.prepare_batch

    def begin(self, *a, **k):
        if site == "solve_begin" and self.batch > 4 and not fired:
            fired.append("begin")
            raise ImcomError(-3, "device workspace: out of memory (injected)")
        return real_begin(self, *a, **k)

    def end(self):
        if site == "solve_end" and self.batch > 4 and not fired:
            fired.append("end")
            real_end(self)  # (the outstanding begin is ended; its results are discarded by the halves)
            raise ImcomError(-3, "device workspace: out of memory (injected)")
        return real_end(self)

    def prepare(cfg_, pool_, tables_, chunk, *a, **k):
        if site == "next_batch" and list(chunk) == plan[1] and not fired:
            fired.append("prepare")  # the second pass, prepared while the first pass's solve is queued
            raise RuntimeError("HIP out of memory (injected)")
        return real_prepare(cfg_, pool_, tables_, chunk, *a, **k)

    with monkeypatch.context() as mp:
        mp.setattr(StampBatch, "solve_begin", begin)
        mp.setattr(StampBatch, "solve_end", end)
        mp.setattr(blockrun, "prepare_batch", prepare)
        got = coadd_block(cfg, pool, tabs, n1P, n_expo, chunks=plan, claim=claim if with_claim else None)
        torch.cuda.synchronize()
    assert fired and got.passes_halved == 1 and sorted(got.chunks_done) == [0, 1]
    assert not with_claim or asked == [0, 1]  # every pass asked for once: the failed preparation did not draw a third
    ref = coadd_block(cfg, pool, tabs, n1P, n_expo, chunks=[todo[:4], todo[4:8], todo[8:]])
    torch.cuda.synchronize()
    assert torch.equal(got.out_map, ref.out_map) and torch.equal(got.T_weightmap, ref.T_weightmap)
    for k in ("UC", "Sigma", "kappa", "Tsum", "Neff"):
        assert torch.equal(got.maps[k], ref.maps[k]), k


def test_plan_is_exact_and_reproducible_under_a_memory_cap():
    """ONE owner for device memory: the library works in a torch tensor (imcom_ctx_set_workspace; _lib.Context), so
    torch's allocator accounts for every byte and ``plan_block`` sizes passes from it alone.  With the free memory capped by a hog
    tensor: (1) the plan made on a fresh process state and the plan made after a block has run (workspace tensor and per-batch buffers
    now exist and count as available) are the SAME plan; (2) the passes are smaller than without the cap; (3) the block runs through
    without a pass being halved and its peak stays inside the cap; (4) the library itself allocated nothing: torch's peak IS the
    device's."""
    import torch

    from pyimcom_amd import synth
    from pyimcom_amd.blockrun import available_bytes, coadd_block, pass_bytes, plan_block, release_buffers
    from pyimcom_amd.select import InStampPool
    from pyimcom_amd.stamps import NB, PSFGroupTables

    cfg = synth.CONFIGS["cfg2"]
    n1P, E = 8, cfg.n_expo
    inst = synth.make_instamps(cfg, n1P, E, np.random.default_rng(8))
    pool = InStampPool(inst, cfg.n_inframe)
    psfs, target = synth.make_psfs(cfg, E)
    tabs = PSFGroupTables(psfs, target, cfg.nfft)
    ctx = tabs.ctx
    release_buffers()
    ctx.release_workspace()
    torch.cuda.empty_cache()
    free_plan = [len(c) for c in plan_block(cfg, pool, tabs, n1P)]
    assert free_plan == [64]  # nothing in the way: the whole block in one pass
    # cap: room for ~24 stamps
    ldm = (cfg.m + NB - 1) // NB * NB
    room = pass_bytes(24, 2304, ldm, 1, "Cholesky", nv=1, n_inframe=cfg.n_inframe) + (3 << 30)
    spare = available_bytes(pool.device, ctx) - room
    hog = torch.empty(spare, dtype=torch.uint8, device=pool.device)
    first = [len(c) for c in plan_block(cfg, pool, tabs, n1P)]
    assert sum(first) == 64 and 8 <= max(first) < 40, first
    free0 = torch.cuda.mem_get_info()[0]
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    maps = coadd_block(cfg, pool, tabs, n1P, E)
    torch.cuda.synchronize()
    assert maps.passes_halved == 0 and maps.chunk_sizes == first
    peak = torch.cuda.max_memory_allocated() - base
    assert peak <= room, (peak, room)
    # (4): what left the driver's free pool is what torch reserved -- no allocation beside torch's
    taken = free0 - torch.cuda.mem_get_info()[0]
    assert taken <= torch.cuda.memory_reserved() - 0 and ctx._ws is not None and ctx.workspace_bytes() == ctx._ws.numel()
    again = [len(c) for c in plan_block(cfg, pool, tabs, n1P)]
    assert again == first, (first, again)
    maps2 = coadd_block(cfg, pool, tabs, n1P, E)
    torch.cuda.synchronize()
    assert maps2.passes_halved == 0 and torch.equal(maps2.out_map, maps.out_map)
    del hog
    release_buffers()
    ctx.release_workspace()
    torch.cuda.empty_cache()


def test_pixels_ordered_by_psf_give_the_same_block(monkeypatch):
    """prepare_batch orders a stamp's input pixels by PSF (PSF group, then exposure) instead of the reference's nine InStamp segments
    (coadd.py:937), so that a tile of the A builder stays on one overlap table.  Everything a block hands out is a sum over the input
    pixels: the maps of the two orders agree to the rounding of those sums (NOT bit for bit: the order of A's rows is the order of the
    factorisation's operations), T comes back in the reference's order, and the permutation is one (a stable sort by PSF index)."""
    import torch

    from pyimcom_amd import synth
    from pyimcom_amd.blockrun import coadd_block, prepare_batch
    from pyimcom_amd.select import InStampPool
    from pyimcom_amd.stamps import BlockTables

    cfg = synth.CONFIGS["small"]
    n1P, E = 4, 4
    inst = synth.make_instamps(cfg, n1P, E, np.random.default_rng(5))
    pool = InStampPool(inst, cfg.n_inframe)
    psfs, target = synth.make_psfs(cfg, E)
    ng = (n1P + 3) // 2
    tabs = BlockTables({(gj, gi): synth.group_psfs(psfs, gj, gi) for gj in range(ng) for gi in range(ng)}, target, cfg.nfft, capacity=2000)
    chunk = [(2, 2), (2, 3), (3, 2)]
    sb = prepare_batch(cfg, pool, tabs, chunk, n1P, E)
    assert sb.perm is not None
    for q in range(len(chunk)):
        n = int(sb.n[q])
        slot = sb.psf[q, :n].cpu().numpy()
        assert np.all(np.diff(slot) >= 0) and len(set(slot.tolist())) > E  # runs of one PSF, several groups
        p = sb.perm[q, :n].cpu().numpy()
        assert sorted(p.tolist()) == list(range(n))
        for s_ in set(slot.tolist()):
            assert np.all(np.diff(p[slot == s_]) > 0)  # stable: inside a PSF the reference's order
    sb.run()
    a = coadd_block(cfg, pool, tabs, n1P, E, batch=5)
    Ta = sb.result().T(1).clone()
    monkeypatch.setenv("IMCOM_PIXEL_ORDER", "segments")
    sb2 = prepare_batch(cfg, pool, tabs, chunk, n1P, E)
    assert sb2.perm is None
    sb2.run()
    b = coadd_block(cfg, pool, tabs, n1P, E, batch=5)
    torch.cuda.synchronize()
    Tb = sb2.result().T(1)
    assert float((Ta - Tb).abs().max()) <= 2e-6 * float(Tb.abs().max())  # the same T, pixel for pixel, in the reference's order
    assert float((a.out_map - b.out_map).abs().max()) <= 1e-5 * float(b.out_map.abs().max())
    for k in ("UC", "Sigma", "kappa", "Tsum", "Neff"):
        x, y = a.maps[k], b.maps[k]
        assert torch.allclose(x, y, rtol=2e-5, atol=1e-7 * float(y.abs().max())), k
    assert torch.allclose(a.T_weightmap, b.T_weightmap, rtol=1e-5, atol=1e-8)
