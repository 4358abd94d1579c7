"""The other LA kernels at block level and at full stamp size (BASELINE configs[2]: "Eigendecomposition kappa-sweep path,
8-exposure stamps, batched across one block"): ``coadd_block`` with ``kernel="Eigen"`` on 48 x 48-output stamps at 8 exposures
with a PSF group per 2 x 2 InStamps, and with ``kernel="Iterative"`` at cfg-2 size in ONE pass of 256 stamps (the blocked-CG
workspace of csrc/iter_block.hip at the batch a block really hands over)."""

import dataclasses

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_eigen_kernel_block_at_full_stamp_size():
    """cfg-3 geometry (N ~ 2.9k, m = 2304, three kappa nodes), a block of 4 x 4 output stamps whose PSFs differ from group to
    group (9 groups, self / cross / input-output tables through the arena): the block maps against the stamp-by-stamp route
    (same device kernels, one stamp per batch), two stamps against orc.eigen_kernel on the device's A and -B/2 (numpy eigh + the
    C lakernel1, with the flip accounting of test_cfg3_full_vs_oracle), and the planner's memory estimate
    stamp_bytes("Eigen") against what the pass really took (torch allocations + the library workspace)."""
    import torch

    from oracle import oracle as orc
    from pyimcom_amd import synth
    from pyimcom_amd.block import BlockMaps
    from pyimcom_amd.blockrun import coadd_block, count_pixels, estimate_pixels, pass_bytes, plan_block, prepare_batch, release_buffers
    from pyimcom_amd.stamps import NB, BlockTables
    from tests.test_gpu_bigblock import _workload

    cfg = synth.CONFIGS["cfg3"]
    n1P, E = 4, cfg.n_expo
    release_buffers()
    torch.cuda.empty_cache()
    inst, pool, target, groups, counts, provider = _workload(cfg, n1P, E, seed=33)
    tabs = BlockTables(groups, target, cfg.nfft, group_count=counts, bulk_provider=provider, cells=True, capacity=4000)
    chunks = plan_block(cfg, pool, tabs, n1P)
    assert sum(len(c) for c in chunks) == n1P * n1P
    tabs.ctx.release_workspace()  # (earlier tests of the session may have left a larger workspace on the shared context: measure this pass's own)
    base = torch.cuda.memory_allocated()  # (the library's workspace is a torch tensor: torch's figures are the whole device)
    assert tabs.ctx.workspace_bytes() == 0
    torch.cuda.reset_peak_memory_stats()
    maps = coadd_block(cfg, pool, tabs, n1P, E, chunks=chunks)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(maps.out_map).all()) and float(maps.out_map.abs().max()) > 0
    # memory: the estimate the planner sizes batches with must cover what the largest pass took
    bmax = max(len(c) for c in chunks)
    took = torch.cuda.max_memory_allocated() - base
    exact = count_pixels(cfg, pool, n1P)  # the planner sizes passes from the selection's own counts
    assert np.abs(estimate_pixels(cfg, pool, n1P) / exact - 1.0).max() < 0.1  # (what a pool without a GPU falls back to)
    ldn = (int(exact.max()) + NB - 1) // NB * NB
    from pyimcom_amd.blockrun import TABLE_WS_BYTES

    est = pass_bytes(bmax, ldn, (cfg.m + NB - 1) // NB * NB, 1, "Eigen", nv=len(cfg.kappaC), n_inframe=cfg.n_inframe, table_ws=TABLE_WS_BYTES)
    print(f"[eigen block] largest pass {bmax} stamps: took {took / 2**30:.2f} GiB, pass_bytes {est / 2**30:.2f} GiB")
    assert took <= est, (took, est)  # the planner fills the available memory by this figure: what a pass takes must be inside it
    assert est <= 1.6 * took  # ... and it is not a wild overestimate either (it counts two resident batches; this block has one pass)

    # stamp by stamp
    one = BlockMaps(n1P, cfg.n2, cfg.fade, cfg.n_inframe, E, ctx=tabs.ctx)
    keep = {}
    for j in range(1, n1P + 1):
        for i in range(1, n1P + 1):
            sb = prepare_batch(cfg, pool, tabs, [(j, i)], n1P, E)
            sb.build()
            sb.solve()
            sb.coadd()
            one.add(sb.results(), [j], [i])
            if (j, i) in ((1, 1), (3, 2)):
                n = int(sb.n[0])
                keep[(j, i)] = (n, sb.A[0, :n, :n].cpu().numpy(), np.ascontiguousarray(sb.Bt[0, :n, : cfg.m].cpu().numpy().T), sb.result())
    torch.cuda.synchronize()
    for k in ("UC", "Sigma", "kappa", "Tsum", "Neff"):
        a, b = maps.maps[k], one.maps[k]
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7 * float(b.abs().max())), k
    a, b = maps.out_map, one.out_map
    assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())
    assert torch.allclose(maps.T_weightmap, one.T_weightmap, rtol=1e-5, atol=1e-8)

    # two stamps against the oracle's EigenKernel
    flips = 0
    for (j, i), (n, A, mB, res) in keep.items():
        assert 2600 < n < 3200 and res.info[0] == 0 and exact[j - 1, i - 1] == n  # (the planner's counts are the selection's)
        T, UC, Sigma, kappa, _ = orc.eigen_kernel(A, mB, float(tabs.C), cfg.kappaC, cfg.uctarget, cfg.sigmamax)
        k_gpu, k_ref = res.kappa[0].cpu().numpy().ravel().astype(np.float64), kappa.astype(np.float64)
        same = np.abs(k_gpu / k_ref - 1.0) <= 1e-6
        flips += int((~same).sum())
        assert np.all(np.abs(UC[~same] - cfg.uctarget) <= 1e-9), "kappa decision flipped away from a tie"
        lam = np.linalg.eigvalsh(A)
        kap = cfg.kappaC[0] * float(tabs.C)
        cond = (lam[-1] + kap) / (max(lam[0], 0.0) + kap)
        Tg = res.T(0, order="batch").cpu().numpy()  # (A and -B/2 were taken in the batch's pixel order)
        assert np.abs(Tg[same] - T[same]).max() <= (1e-6 + 50 * cond * 2.2e-16) * np.abs(T).max(), (j, i)
        assert np.allclose(res.UC[0].cpu().numpy().ravel()[same], UC[same], rtol=1e-5 + 50 * cond * 2.2e-16, atol=1e-9)
        assert np.allclose(res.Sigma[0].cpu().numpy().ravel()[same], Sigma[same], rtol=1e-5 + 50 * cond * 2.2e-16, atol=1e-9)
        # ... and inside the block's maps (fade 0: the stamp's tile is the maps' window)
        ys, xs = slice((j - 1) * cfg.n2, j * cfg.n2), slice((i - 1) * cfg.n2, i * cfg.n2)
        assert torch.equal(maps.maps["kappa"][0, ys, xs].cpu(), res.kappa[0].cpu()) or torch.allclose(maps.maps["kappa"][0, ys, xs].cpu(), res.kappa[0].cpu(), rtol=1e-6)
    assert flips <= 2
    release_buffers()


def test_iterative_kernel_block_one_pass_of_256_stamps():
    """cfg-2 geometry with kernel = "Iterative" (lakernel.py:533-744; coadd.py:1104-1107 clamp): a 16 x 16-stamp block as ONE
    pass of 256 stamps -- 256 x 144 patches of 16 output pixels, each with its dense union sub-matrix: the blocked-CG workspace is
    sub-batched inside the library -- against the same block in passes of 48 stamps (bit for bit: a patch's recurrences do not
    depend on its neighbours in the launch), and two stamps against orc.iter_kernel on the device's A and -B/2 at the tolerances
    of the Iterative kernel (tests/parity.py TOL_ITER)."""
    import torch

    from oracle import oracle as orc
    from pyimcom_amd import synth
    from pyimcom_amd.blockrun import coadd_block, prepare_batch, release_buffers
    from pyimcom_amd.select import InStampPool
    from pyimcom_amd.stamps import PSFGroupTables
    from tests.parity import TOL_ITER

    cfg = dataclasses.replace(synth.CONFIGS["cfg2"], name="cfg2_iter", kernel="Iterative")
    n1P, E = 16, cfg.n_expo
    release_buffers()
    torch.cuda.empty_cache()
    inst = synth.make_instamps(cfg, n1P, E, np.random.default_rng(77))
    pool = InStampPool(inst, cfg.n_inframe)
    psfs, target = synth.make_psfs(cfg, E)
    tabs = PSFGroupTables(psfs, target, cfg.nfft)
    whole = coadd_block(cfg, pool, tabs, n1P, E, batch=256)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(whole.out_map).all()) and float(whole.out_map.abs().max()) > 0
    assert whole.chunks_done == [0]
    parts = coadd_block(cfg, pool, tabs, n1P, E, batch=48)
    torch.cuda.synchronize()
    assert len(parts.chunks_done) == 6
    assert torch.equal(whole.out_map, parts.out_map) and torch.equal(whole.T_weightmap, parts.T_weightmap)
    for k in ("UC", "Sigma", "kappa", "Tsum", "Neff"):
        assert torch.equal(whole.maps[k], parts.maps[k]), k
    assert float(whole.maps["UC"].min()) >= 1e-32 and float(whole.maps["Sigma"].min()) >= 1e-32  # the clamp of coadd.py:1104-1107

    # two stamps against the oracle, at a kappa where CG converges inside its 30 steps (at kappa / C = 6e-4 it returns an unconverged
    # iterate and WHICH one is a matter of rounding: DESIGN "Iterative kernel and rounding"; the golden tests cover that regime)
    cfg = dataclasses.replace(cfg, kappaC=(0.2,))
    sample = [(1, 1), (9, 12)]
    sb = prepare_batch(cfg, pool, tabs, sample, n1P, E)
    sb.build()
    sb.solve()
    sb.coadd()
    torch.cuda.synchronize()
    res = sb.result()
    g = np.arange(cfg.n2f, dtype=np.float64)
    for q, (j, i) in enumerate(sb.chunk):
        n = int(sb.n[q])
        A = sb.A[q, :n, :n].cpu().numpy()
        mB = np.ascontiguousarray(sb.Bt[q, :n, : cfg.m].cpu().numpy().T)
        oy = np.repeat(float(sb.out_y0[q]) + g, cfg.n2f)
        ox = np.tile(float(sb.out_x0[q]) + g, cfg.n2f)
        T, UC, Sigma, kappa, _ = orc.iter_kernel(A, mB, float(tabs.C), cfg.kappaC, cfg.uctarget, cfg.sigmamax, oy, ox, sb.y[q, :n].cpu().numpy(),
                                                 sb.x[q, :n].cpu().numpy(), float(cfg.rho))
        UC, Sigma = orc.iterative_clamp(UC, Sigma)
        Tg = res.T(q, order="batch").cpu().numpy()
        assert np.array_equal(Tg == 0, T == 0), (j, i)  # the same acceptance discs
        assert np.abs(Tg - T).max() <= TOL_ITER["T"] * np.abs(T).max(), (j, i)
        assert np.allclose(res.Sigma[q].cpu().numpy().ravel(), Sigma, rtol=TOL_ITER["map_rtol"], atol=TOL_ITER["map_atol"])
        assert np.allclose(res.UC[q].cpu().numpy().ravel(), UC, rtol=TOL_ITER["map_rtol"], atol=TOL_ITER["map_atol"])
    release_buffers()
