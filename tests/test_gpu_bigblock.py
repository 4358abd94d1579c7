"""The Block seam at the reference's own block geometry (reference src/pyimcom/coadd.py:2003-2084 with one PSFGrp per 2 x 2
InStamps, psfutil.py:1803-1824): n1P = 84 output stamps per side as configs/paper4_configs/H158_Chol_benchmark.json:28-34
(OUTSIZE [80, 32, ...], PAD 2, FADE 3), and cfg-2 blocks of 48 x 48 stamps at 8 and 10 exposures ("one block =
48 x 48 stamps"; BASELINE configs[3] depth).  Such a block has 1849 (625) PSF groups and 310 k (100-250 k) overlap tables --
390 GB if they were all resident -- so these tests exercise what the small blocks of test_gpu_blockrun.py cannot: arena
indices beyond 2^31 table elements, batches planned as 2-D tiles of cells against the arena, least-recently-used eviction of
table sets, the spectra arena running out of rows, and windowed cross tables -- at full size, as size-independent properties:

* the SAME batches through two arena regimes -- (a) an arena a third of the device memory, whole tables; (b) an arena barely
  larger than one batch's sets, filled with NaN first, cross tables computed only inside the window the separations of two
  grid cells can reach, a spectra arena of a few batches -- must give the block maps BIT FOR BIT: a table read from a slot
  that was evicted, not yet rewritten, or outside its computed window would show as NaN or as a different number;
* the block in small explicit batches agrees to rounding (not bit for bit: the factorisation deals the K loop of launches with
  fewer than 256 tiles to several workgroups, csrc/api.hip splitk_parts, so the summation order of the last block rows
  depends on the batch size);
* a sample of stamps against the oracle's CholKernel on the device's own A and -B/2 (the builders are compared with the oracle
  at this stamp size in test_gpu_fullsize.py).
"""

import dataclasses

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _workload(cfg, n1P, E, seed):
    """InStamp pool of the block and a bulk provider of per-group sampled PSFs (a smooth modulation of the analytic PSFs that
    depends on the group: every group has PSFs of its own, none is stored)."""
    import torch

    from pyimcom_amd import synth
    from pyimcom_amd.select import InStampPool

    inst = synth.make_instamps(cfg, n1P, E, np.random.default_rng(seed))
    pool = InStampPool(inst, cfg.n_inframe)
    psfs, target = synth.make_psfs(cfg, E)
    base = torch.as_tensor(psfs, device="cuda:0")
    ns = psfs.shape[-1]
    lin = torch.arange(ns, dtype=torch.float64, device="cuda:0") - ns // 2
    ng = (n1P + 3) // 2
    groups = {(gj, gi): None for gj in range(ng) for gi in range(ng)}
    counts = {k: E for k in groups}

    def provider(keys):
        out = torch.empty((len(keys), E, ns, ns), dtype=torch.float64, device="cuda:0")
        for q, (gj, gi) in enumerate(keys):
            mod = 1.0 + 0.02 * torch.sin(0.05 * lin * (1 + gi % 3))[None, None, :] + 0.02 * torch.cos(0.04 * lin * (1 + gj % 5))[None, :, None]
            p = base * mod * (1.0 + 1e-3 * ((7 * gj + 3 * gi) % 11))
            out[q] = p / p.sum(dim=(1, 2), keepdim=True)
        return out.reshape(-1, ns, ns)

    return inst, pool, target, groups, counts, provider


def _equal_maps(a, b):
    import torch

    assert torch.equal(a.out_map, b.out_map)
    assert torch.equal(a.T_weightmap, b.T_weightmap)
    for k in ("UC", "Sigma", "kappa", "Tsum", "Neff"):
        assert torch.equal(a.maps[k], b.maps[k]), k


@pytest.mark.parametrize("name,n1P,E", [("paper4_n1P84_E6", 84, 6), ("cfg2_n1P48_E8", 48, 8), ("cfg2_n1P48_E10", 48, 10)])
def test_block_of_reference_size(name, n1P, E):
    import torch

    from oracle import oracle as orc
    from pyimcom_amd import synth
    from pyimcom_amd.blockrun import chunk_keys, coadd_block, plan_block, prepare_batch, stamp_groups
    from pyimcom_amd.stamps import BlockTables

    if n1P == 84:  # the reference's production block: 32 x 32-output stamps at 0.039", fade 3, six input layers
        cfg = dataclasses.replace(synth.CONFIGS["cfg2"], name=name, n2=32, fade=3, dtheta_as=0.0390625, n_expo=E, n_inframe=2)
    else:
        cfg = dataclasses.replace(synth.CONFIGS["cfg2"], name=name, n_expo=E)
    nst = n1P + 2
    torch.cuda.empty_cache()  # buffers cached by earlier tests: the arenas below take fixed shares of what is free
    inst, pool, target, groups, counts, provider = _workload(cfg, n1P, E, seed=84)
    ng2 = (cfg.nsamp + 12) ** 2

    # (a) roomy arena, whole tables
    big = BlockTables(groups, target, cfg.nfft, group_count=counts, bulk_provider=provider)
    assert big.capacity * ng2 > 2**31, "the arena must reach beyond 31-bit element offsets for this test to mean anything"
    assert big.block_demand() > (300_000 if n1P == 84 else 90_000)
    chunks = plan_block(cfg, pool, big, n1P)  # tiles of 2 x 2-stamp cells
    assert sorted(t for c in chunks for t in c) == [(j, i) for j in range(1, n1P + 1) for i in range(1, n1P + 1)]
    assert 100 <= min(len(c) for c in chunks) and max(len(c) for c in chunks) <= 256  # e.g. 42 tiles of 6 x 7 cells = 168 stamps x 12 workgroups: 3.94 rounds of 512
    ref = coadd_block(cfg, pool, big, n1P, E, chunks=chunks)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(ref.out_map).all()) and float(ref.out_map.abs().max()) > 0
    assert big.evictions == 0 or big.capacity < big.block_demand()
    need = max(big.demand(chunk_keys(c, nst)) for c in chunks)
    groups_per_batch = max(len({g for t in c for g in stamp_groups(t[0], t[1], nst)}) for c in chunks)
    del big
    torch.cuda.empty_cache()

    # (b) tight arena: a bit more than the largest batch needs; NaN everywhere; windowed cross tables; few spectra rows
    tight = BlockTables(groups, target, cfg.nfft, group_count=counts, bulk_provider=provider, capacity=int(1.3 * need), cells=True,
                        spec_capacity=1 + int(2.5 * groups_per_batch) * E)
    tight.tables[1:].fill_(float("nan"))
    got = coadd_block(cfg, pool, tight, n1P, E, chunks=chunks)
    torch.cuda.synchronize()
    assert tight.evictions >= len(chunks) // 2 and tight.evicted_tables > need and tight.spectra_resets >= 1
    assert tight.computed_tables < 1.25 * sum(tight.demand(chunk_keys(c, nst)) for c in chunks)  # neighbours' shared sets mostly survive
    assert bool(torch.isnan(tight.tables[1 : tight.used + 1]).any()), "nothing was pruned"
    _equal_maps(got, ref)

    # small explicit batches (cell order cut every 60 stamps): equal to rounding
    small = coadd_block(cfg, pool, tight, n1P, E, batch=60)
    torch.cuda.synchronize()
    a, b = ref.out_map, small.out_map
    assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max())
    for k in ("UC", "Sigma", "kappa", "Tsum", "Neff"):
        x, y = ref.maps[k], small.maps[k]
        assert torch.allclose(x, y, rtol=1e-4, atol=1e-7 * float(x.abs().max())), k

    # a sample of stamps (a corner, an edge, the middle, across tile boundaries) against the oracle's Cholesky kernel
    sample = [(1, 1), (n1P, n1P // 2), (n1P // 2, n1P // 2 + 1), (17, 16)]
    sb = prepare_batch(cfg, pool, tight, sample, n1P, E)
    sb.build()
    sb.solve()
    sb.coadd()
    torch.cuda.synchronize()
    res = sb.result()
    for q, (j, i) in enumerate(sb.chunk):
        n = int(sb.n[q])
        A = sb.A[q, :n, :n].cpu().numpy()
        mB = np.ascontiguousarray(sb.Bt[q, :n, : cfg.m].cpu().numpy().T)
        assert np.array_equal(A, A.T) and n > 1500
        To, Uo, So, ko, _ = orc.chol_kernel(A, mB, float(tight.C), np.array(cfg.kappaC), cfg.uctarget, cfg.sigmamax)
        lam = np.linalg.eigvalsh(A)
        kap = cfg.kappaC[0] * tight.C
        cond = (lam[-1] + kap) / (max(lam[0], 0.0) + kap)
        f2, s2 = 2 * cfg.fade, (cfg.n2f, cfg.n2f)
        inner = (slice(f2, cfg.n2f - f2),) * 2  # the tapers of coadd.py:1118-1122, 1320-1327 leave these output pixels untouched
        T = res.T(q, order="batch").cpu().numpy().reshape(cfg.n2f, cfg.n2f, n)[inner]  # (A, -B/2 above are in the batch's pixel order)
        assert np.abs(T - To.reshape(cfg.n2f, cfg.n2f, n)[inner]).max() <= (1e-6 + 50 * cond * 2.2e-16) * np.abs(To).max(), (j, i)
        assert np.allclose(res.kappa[q].cpu().numpy()[inner], ko.reshape(s2)[inner], rtol=1e-5, atol=0)
        assert np.allclose(res.UC[q].cpu().numpy()[inner], Uo.reshape(s2)[inner], rtol=1e-5 + 50 * cond * 2.2e-16, atol=1e-9)
        assert np.allclose(res.Sigma[q].cpu().numpy()[inner], So.reshape(s2)[inner], rtol=1e-5 + 50 * cond * 2.2e-16, atol=1e-9)
        # and the block maps hold this stamp's inner pixels (no neighbour overlaps there) as this batch of four produced them
        y0, x0 = (j - 1) * cfg.n2, (i - 1) * cfg.n2
        for k, mine in (("kappa", res.kappa), ("UC", res.UC), ("Sigma", res.Sigma)):
            blk = ref.maps[k][0, y0 + f2 : y0 + cfg.n2f - f2, x0 + f2 : x0 + cfg.n2f - f2]
            assert torch.allclose(blk, mine[q][inner], rtol=1e-4, atol=1e-9), (k, j, i)


def test_bench_block_leg_workload_with_identical_groups_equals_one_group():
    """bench.py's block leg, checked numerically: its workload generator with the SAME PSF images and unrotated sampling
    positions in every 2 x 2 group, through the leg's own route (resident PSF images -> bulk sampling -> spectra -> per-group
    self / cross / input-output table sets -> pair maps -> tiles of cells), must reproduce the block coadded with ONE PSF
    group (PSFGroupTables of the same sampled PSFs; stamps against the oracle at this size: test_gpu_fullsize.py) -- to the
    conditioning of the solve, since the tables of a pair come from different FFT evaluations on the two routes."""
    import torch

    import bench
    from pyimcom_amd import psfs as psfmod
    from pyimcom_amd.blockrun import coadd_block
    from pyimcom_amd.stamps import BlockTables, PSFGroupTables

    dev = "cuda:0"
    n1P = 16
    torch.cuda.empty_cache()
    cfg, inst, pool, psfs, target, groups, counts, img_all, yxco_all = bench.block_workload(dev, n1P, identical=True)
    E, ns = cfg.n_expo, psfs.shape[-1]
    order = {k: q for q, k in enumerate(groups)}

    def sample_groups(keys):
        idx = torch.as_tensor([order[k] for k in keys], device=dev)
        return psfmod.sample_psf(img_all[idx].reshape(-1, ns + 16, ns + 16), ns, yxco_all[idx].reshape(-1, 2, ns, ns), psf_norm=True)

    tabs = BlockTables(groups, target, cfg.nfft, group_count=counts, bulk_provider=sample_groups, cells=True)
    grp = coadd_block(cfg, pool, tabs, n1P, E)
    sampled = sample_groups([(0, 0)])
    one = coadd_block(cfg, pool, PSFGroupTables(sampled, target, cfg.nfft), n1P, E)
    torch.cuda.synchronize()
    a, b = one.out_map, grp.out_map
    assert bool(torch.isfinite(a).all()) and float(a.abs().max()) > 0
    assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
    for k in ("UC", "Sigma", "kappa", "Tsum", "Neff"):
        x, y = one.maps[k], grp.maps[k]
        assert torch.allclose(x, y, rtol=1e-4, atol=1e-7 * float(x.abs().max())), k
    assert torch.allclose(one.T_weightmap, grp.T_weightmap, rtol=1e-5, atol=1e-8)
    # and with the leg's real workload (a PSF of its own per group) the block differs: the groups matter
    cfg2, _, pool2, _, target2, groups2, counts2, img2, yx2 = bench.block_workload(dev, n1P)
    order2 = {k: q for q, k in enumerate(groups2)}

    def sample2(keys):
        idx = torch.as_tensor([order2[k] for k in keys], device=dev)
        return psfmod.sample_psf(img2[idx].reshape(-1, ns + 16, ns + 16), ns, yx2[idx].reshape(-1, 2, ns, ns), psf_norm=True)

    var = coadd_block(cfg2, pool2, BlockTables(groups2, target2, cfg2.nfft, group_count=counts2, bulk_provider=sample2, cells=True), n1P, E)
    assert float((var.out_map - a).abs().max()) > 1e-3 * float(a.abs().max())


def test_eager_groups_leave_no_trace():
    """BlockTables(eager_groups=True): every PSF group of the block is sampled / transformed at the block's start instead of pass by pass
    (and, with few spectra rows, only as many as the arena holds).  Spectra rows and table slots are then handed out in another order --
    the block maps must be the same bits, for the whole block and for a window of stamps."""
    import torch

    import bench
    from pyimcom_amd import psfs as psfmod
    from pyimcom_amd.blockrun import coadd_block
    from pyimcom_amd.stamps import BlockTables

    dev = "cuda:0"
    n1P = 8
    torch.cuda.empty_cache()
    cfg, inst, pool, psfs, target, groups, counts, img_all, yxco_all = bench.block_workload(dev, n1P)
    E, ns = cfg.n_expo, psfs.shape[-1]
    order = {k: q for q, k in enumerate(groups)}
    calls = []

    def sample_groups(keys):
        calls.append(len(keys))
        idx = torch.as_tensor([order[k] for k in keys], device=dev)
        return psfmod.sample_psf(img_all[idx].reshape(-1, ns + 16, ns + 16), ns, yxco_all[idx].reshape(-1, 2, ns, ns), psf_norm=True)

    def block(eager, stamps=None, spec_capacity=None):
        calls.clear()
        tabs = BlockTables(groups, target, cfg.nfft, group_count=counts, bulk_provider=sample_groups, cells=True, eager_groups=eager,
                           spec_capacity=spec_capacity)
        maps = coadd_block(cfg, pool, tabs, n1P, E, batch=16, stamps=stamps)
        torch.cuda.synchronize()
        return maps, list(calls)

    ref, c0 = block(False)
    eag, c1 = block(True)
    assert c1[0] == len(groups) and len(c1) == 1 and len(c0) > 1  # one call for the whole block against one per pass
    few, _ = block(True, spec_capacity=1 + 10 * E)  # the arena holds ten groups: the rest is materialised when first needed
    win = [(j, i) for j in range(3, 7) for i in range(3, 7)]
    ref_w, _ = block(False, stamps=win)
    eag_w, c2 = block(True, stamps=win)
    assert c2[0] == 9  # the window's stamps touch 3 x 3 groups
    for a, b in ((ref, eag), (ref, few), (ref_w, eag_w)):
        assert torch.equal(a.out_map, b.out_map) and torch.equal(a.T_weightmap, b.T_weightmap)
        for k in ("UC", "Sigma", "kappa", "Tsum", "Neff"):
            assert torch.equal(a.maps[k], b.maps[k]), k
