"""Block seam (pyimcom_amd.refblock): the reference's Block containers in, its block maps out.

The containers are rebuilt from tests/golden/stamp_chain*.npz exactly as tests/golden/make_golden_chain.py built them for the
reference's own run (Block / Config / InStamp / InImage as plain objects: pixel arrays, a fixed PSF image and an affine
pixel map per exposure), so the adapter is exercised on what `Block.coadd_output_stamps` would hand it, and its output
for the golden's output stamp is compared with what the reference's chain produced."""

import numpy as np
import pytest


class Empty:
    pass


ARCSEC = np.pi / 180.0 / 3600.0


def reference_block(g, kernel="Cholesky", kappaC=None):
    """blk and the PSFGrp class attributes of make_golden_chain.py, from the arrays it recorded."""
    n1P, n2, fade, n_inimage, n_inframe = (int(v) for v in g["pars"])
    npixpsf, oversamp, dtheta_as = int(g["npixpsf"]), int(g["oversamp"]), float(g["dtheta_as"])
    psfgrp = Empty()  # PSFGrp.setup (psfutil.py:568-613)
    psfgrp.npixpsf, psfgrp.oversamp = npixpsf, oversamp
    psfgrp.nsamp = npixpsf * oversamp - 1
    psfgrp.nfft = npixpsf * oversamp * 2
    psfgrp.dscale = (0.11 * ARCSEC / ARCSEC) / oversamp / ((dtheta_as / 3600.0) * 3600)
    cfg = Empty()
    cfg.n1P, cfg.n2, cfg.fade_kernel, cfg.n2f, cfg.n_inframe = n1P, n2, fade, n2 + 2 * fade, n_inframe
    cfg.dtheta, cfg.instamp_pad = dtheta_as / 3600.0, float(g["instamp_pad_as"]) * ARCSEC
    cfg.linear_algebra, cfg.no_qlt_ctrl = kernel, False
    cfg.kappaC_arr, cfg.uctarget, cfg.sigmamax = (g["kappaC"] if kappaC is None else np.asarray(kappaC)), 1e-6, 0.5
    cfg.psf_circ, cfg.psf_norm, cfg.amp_penalty = True, True, [0.0, 0.0]
    cfg.n_out, cfg.outpsf, cfg.sigmatarget, cfg.use_filter = 1, "GAUSSIAN", 1.1, 2
    cfg.outpsf_extra, cfg.sigmatarget_extra = [], []
    cfg.postage_pad, cfg.psfsplit, cfg.psf_interp = 0, None, "D5512"
    cfg.iter_rtol, cfg.iter_max = 1.5e-3, 30
    blk = Empty()
    blk.cfg, blk.n_inimage, blk.pad_sides = cfg, n_inimage, ""
    blk.outwcs = Empty()
    blk.outwcs.all_pix2world = lambda arr, origin: np.asarray(arr, dtype=np.float64)
    blk.inimages = []
    for e in range(n_inimage):
        im = Empty()
        psf, M, t0 = g[f"inpsf{e}"], g[f"inM{e}"], g[f"int0{e}"]
        im.get_psf_pos = (lambda p: (lambda point, use_shortrange=True: p))(psf)
        im.outpix2world2inpix = (lambda M_, t_: (lambda xy: np.asarray(xy) @ M_.T + t_))(M, t0)
        blk.inimages.append(im)
    nst = n1P + 2
    blk.instamps = [[None] * nst for _ in range(nst)]
    for j in range(nst):
        for i in range(nst):
            st = Empty()
            st.x_val, st.y_val, st.data = g[f"in{j}{i}_x"], g[f"in{j}{i}_y"], g[f"in{j}{i}_data"]
            st.pix_cumsum = g[f"in{j}{i}_cum"]
            st.pix_count = np.diff(st.pix_cumsum.astype(np.int64)).astype(np.uint32)
            if j % 2 == 0 and i % 2 == 0:
                st.psf_compute_point_pix = [i * n2 - 0.5, j * n2 - 0.5]  # coadd.py:710-714
            blk.instamps[j][i] = st
    return blk, psfgrp


def test_unsupported_configurations_fail_loudly():
    """PSFINTERP G4460 and PSF splitting have no device path: IMCOM_ERR_UNSUPPORTED, not another interpolator (no GPU needed)."""
    from pyimcom_amd._lib import ImcomError
    from pyimcom_amd.refblock import IMCOM_ERR_UNSUPPORTED, check_supported

    cfg = Empty()
    cfg.linear_algebra = "Cholesky"
    check_supported(cfg)
    for attr, val in (("psf_interp", "G4460"), ("psf_interp", "g4460"), ("psfsplit", [3.0, 6.0, 1e-3]), ("linear_algebra", "QR")):
        bad = Empty()
        bad.linear_algebra = "Cholesky"
        setattr(bad, attr, val)
        with pytest.raises(ImcomError) as ei:
            check_supported(bad)
        assert ei.value.status == IMCOM_ERR_UNSUPPORTED


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["stamp_chain", "stamp_chain_mid"])
def test_block_seam_vs_reference_golden(golden, name):
    from pyimcom_amd.refblock import coadd_output_stamps

    g = golden(name)
    blk, psfgrp = reference_block(g)
    j_st, i_st = int(g["j_st"]), int(g["i_st"])
    coadd_output_stamps(blk, psfgrp, flat_penalty=float(g["flat_penalty"]), stamps=[(j_st, i_st)], finalize=False, batch=4)
    n2, n2f = blk.cfg.n2, blk.cfg.n2f
    ys, xs = slice((j_st - 1) * n2, (j_st - 1) * n2 + n2f), slice((i_st - 1) * n2, (i_st - 1) * n2 + n2f)
    lam = np.linalg.eigvalsh(g["A"])
    kap = float(g["kappaC"][0]) * float(g["C"][0])
    cond = (lam[-1] + kap) / (max(lam[0], 0.0) + kap)
    ref_img = g["outimage"]
    nside = blk.cfg.n1P * n2 + 2 * blk.cfg.fade_kernel
    assert blk.out_map.shape == (1, blk.cfg.n_inframe, nside, nside)
    assert np.abs(blk.out_map[:, :, ys, xs] - ref_img).max() <= 2e-5 * np.abs(ref_img).max()
    for key, arr in (("UC", blk.UC_map), ("Sigma", blk.Sigma_map), ("kappa", blk.kappa_map)):
        assert np.allclose(arr[:, ys, xs], g[key], rtol=1e-5 + 50 * cond * 2.2e-16, atol=1e-9), key
    assert np.allclose(blk.Tsum_map[:, ys, xs], g["Tsum_inpix"], rtol=2e-5, atol=2e-5 * np.abs(g["Tsum_inpix"]).max())
    assert np.allclose(blk.Neff_map[:, ys, xs], g["Neff"], rtol=1e-3, atol=0)
    assert np.allclose(blk.T_weightmap[:, :, j_st - 1, i_st - 1], g["Tsum_stamp"], rtol=2e-5, atol=2e-5 * np.abs(g["Tsum_stamp"]).max())
    # nothing outside the stamp's footprint was touched
    mask = np.ones(blk.UC_map.shape[-2:], bool)
    mask[ys, xs] = False
    assert np.all(blk.UC_map[0][mask] == 0) and np.all(blk.out_map[0, 0][mask] == 0)


def oracle_block(g, kernel, kC, flat_penalty, no_qlt_ctrl=False):
    """The reference's stamp loop over the WHOLE block of tests/golden/stamp_chain*.npz with the oracle (coadd.py:2003-2084:
    per stamp _process_input_stamps -> system matrices from the PSFOvl of its groups -> LA kernel -> map tapers ->
    _perform_coaddition -> _output_stamp_wrapper accumulation; then the boundary recovery of 2163-2181), from the raw inputs:
    PSF images, affine pixel maps, InStamp pixels.  Returns the block arrays as the reference names them."""
    from oracle import oracle as orc
    from pyimcom_amd.blockrun import stamp_neighbours
    from tests.test_oracle import _chain_inputs

    geo, inst, group_psfs, group_expo, (n1P, n2, fade, n_inimage, n_inframe) = _chain_inputs(g)
    geo.flat_penalty = flat_penalty
    ns, nst, n2f = geo.nsamp, n1P + 2, n2 + 2 * fade
    tgt = orc.sample_psf(orc.get_outpsf("GAUSSIAN", 1.1, 2, ns, geo.oversamp), ns, None)[None]
    rft_out = orc.pad_and_rfft2(orc.finish_psf_group(tgt, True, True), geo)
    C = float(orc.overlap_out_C(rft_out, geo)[0])
    rft_in = {k: orc.pad_and_rfft2(v, geo) for k, v in group_psfs.items()}
    rho = float(g["instamp_pad_as"]) / float(g["dtheta_as"])
    kC = np.asarray(kC, dtype=np.float64)
    nside = n1P * n2 + 2 * fade
    out = {k: np.zeros((1, nside, nside), np.float32) for k in ("UC", "Sigma", "kappa", "Tsum", "Neff")}
    out_map = np.zeros((1, n_inframe, nside, nside), np.float32)
    Tw = np.zeros((1, n_inimage, n1P, n1P), np.float32)
    g1 = np.arange(n2f, dtype=np.float64)
    for j in range(1, n1P + 1):
        for i in range(1, n1P + 1):
            ids, pvx, pvy = stamp_neighbours(j, i, n2, nst)
            piv = [(None if np.isnan(a) else a, None if np.isnan(b) else b) for a, b in zip(pvx, pvy)]
            nine = [inst[divmod(int(k), nst)] if k >= 0 else None for k in ids]
            sels = [None if t is None else orc.select_pixels(t[0], t[1], pv, rho) for t, pv in zip(nine, piv)]
            groups = [None if k < 0 else (int(k) // nst >> 1, int(k) % nst >> 1) for k in ids]
            x, y, indata, expo, cum = orc.process_input_stamps(nine, piv, rho)
            ox, oy = (i - 1) * n2 - fade + g1, (j - 1) * n2 - fade + g1
            if no_qlt_ctrl:  # coadd.py:1020-1025: no system matrices at all
                A, mB = None, np.zeros((n2f * n2f, x.size))
            else:
                A, mB = orc.stamp_system_groups(nine, sels, groups, rft_in, rft_out, geo, ox, oy, group_expo)
            oyy, oxx = np.meshgrid(oy, ox, indexing="ij")
            if kernel == "Cholesky":
                T, UC, Sg, kp, _ = orc.chol_kernel(A, mB, C, kC, 1e-6, 0.5)
            elif kernel == "Eigen":
                T, UC, Sg, kp, _ = orc.eigen_kernel(A, mB, C, kC, 1e-6, 0.5)
            elif kernel == "Iterative":
                T, UC, Sg, kp, _ = orc.iter_kernel(A, mB, C, kC, 1e-6, 0.5, oyy.ravel(), oxx.ravel(), y, x, rho)
                UC, Sg = orc.iterative_clamp(UC, Sg)  # coadd.py:1104-1107
            else:
                T, UC, Sg, kp, _ = orc.empir_kernel(A, mB, C, kC, oyy.ravel(), oxx.ravel(), y, x, rho, no_qlt_ctrl=no_qlt_ctrl)
            s2 = (n2f, n2f)
            UC, Sg, kp = (np.array(v, dtype=np.float32).reshape(s2).copy() for v in (UC, Sg, kp))
            if not no_qlt_ctrl:  # (_build_system_matrices returns before the tapers in that mode)
                for a in (kp, Sg, UC):  # coadd.py:1118-1122
                    orc.trapezoid(a, fade)
            outimage, Tst, Tin, Neff = orc.perform_coaddition(T[None].copy(), indata, expo, n_inimage, n2f, n2, fade, cum)
            orc.block_accumulate(out_map, outimage, j, i, n2, fade)
            for name, v in (("UC", UC), ("Sigma", Sg), ("kappa", kp), ("Tsum", Tin[0]), ("Neff", Neff[0])):
                orc.block_accumulate(out[name], np.asarray(v, dtype=np.float32)[None], j, i, n2, fade)
            Tw[0, :, j - 1, i - 1] = Tst[0]
    orc.trapezoid_recover(out_map, fade)
    for v in out.values():
        orc.trapezoid_recover(v, fade)
    return {"out_map": out_map, "UC_map": out["UC"], "Sigma_map": out["Sigma"], "kappa_map": out["kappa"], "Tsum_map": out["Tsum"],
            "Neff_map": out["Neff"], "T_weightmap": Tw}


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["stamp_chain", "stamp_chain_mid"])
@pytest.mark.parametrize("kernel,kC", [("Cholesky", [2e-3]), ("Cholesky", [1e-4, 1e-2, 1e-1]), ("Eigen", [2e-3]), ("Eigen", [1e-4, 1e-1]),
                                       ("Iterative", [3e-2]), ("Empirical", [2e-3]), ("Empirical-nqc", [2e-3])])
def test_block_seam_whole_block_vs_oracle(golden, name, kernel, kC):
    """All n1P x n1P stamps of the reference chain's block, boundary recovery included, through ``coadd_output_stamps`` with
    each of the four LA kernels (Cholesky and Eigen with one and with several kappa nodes), against the oracle's restatement
    of the reference's stamp loop on the same containers (coadd.py:1939-2084, 2163-2181).  Every block array is compared, and
    the stamp the reference itself ran (the golden's) is inside the block."""
    from pyimcom_amd.refblock import coadd_output_stamps

    g = golden(name)
    fp = float(g["flat_penalty"])
    nqc = kernel.endswith("-nqc")  # EMPIRNQC: the Empirical kernel without quality control (coadd.py:856-858, 1020-1025; lakernel.py:774-777)
    kernel = kernel.split("-")[0]
    blk, psfgrp = reference_block(g, kernel, kC)
    blk.cfg.no_qlt_ctrl = nqc
    maps = coadd_output_stamps(blk, psfgrp, flat_penalty=fp, batch=3)
    ref = oracle_block(g, kernel, kC, fp, nqc)
    if nqc:
        assert not blk.UC_map.any() and not blk.Sigma_map.any() and not blk.kappa_map.any() and np.abs(blk.out_map).max() > 0
    assert blk.out_map.shape == (1, blk.cfg.n_inframe, maps.nside, maps.nside) and blk.T_weightmap.shape == (1, blk.n_inimage, blk.cfg.n1P, blk.cfg.n1P)
    cg = kernel == "Iterative"  # CG at rtol 1.5e-3: the iterate the loop stops at carries the rounding of its dot products
    rt = 2e-3 if cg else 5e-5
    a, b = blk.out_map, ref["out_map"]
    assert np.isfinite(a).all() and np.abs(b).max() > 0
    assert np.abs(a - b).max() <= (3e-4 if cg else 5e-5) * np.abs(b).max(), np.abs(a - b).max() / np.abs(b).max()
    for k in ("UC_map", "Sigma_map", "kappa_map", "Tsum_map", "Neff_map", "T_weightmap"):
        a, b = getattr(blk, k), ref[k]
        assert a.shape == b.shape and np.allclose(a, b, rtol=rt, atol=(2e-5 if cg else 2e-6) * np.abs(b).max()), (k, np.abs(a - b).max(), np.abs(b).max())
    if kernel == "Iterative":
        assert blk.UC_map[blk.UC_map != 0].min() > 0 and blk.Sigma_map.min() >= 0
    # the same block stamp by stamp (explicit list, one stamp per pass) gives the same maps
    if kernel == "Cholesky" and len(kC) == 1:
        blk2, _ = reference_block(g, kernel, kC)
        n1P = blk.cfg.n1P
        coadd_output_stamps(blk2, psfgrp, flat_penalty=fp, batch=1, stamps=[(j, i) for i in range(1, n1P + 1) for j in range(1, n1P + 1)])
        for k in ref:
            assert np.allclose(getattr(blk2, k), getattr(blk, k), rtol=1e-5, atol=1e-6 * np.abs(ref[k]).max()), k


@pytest.mark.gpu
def test_block_seam_at_reference_block_size():
    """``coadd_output_stamps`` on a duck-typed Block of the reference's production geometry (n1P = 84, 32 x 32-output stamps,
    fade 3, six exposures; configs/paper4_configs/H158_Chol_benchmark.json:28-34): 7396 InStamp objects, 1849 PSF groups whose
    PSF images depend on the group's computation point and are fetched, uploaded and sampled on demand.  The adapter's maps must
    equal, bit for bit, those of ``coadd_block`` driven directly with a bulk provider that samples the same images."""
    import dataclasses

    import torch

    from pyimcom_amd import psfs, synth
    from pyimcom_amd.blockrun import coadd_block, plan_block
    from pyimcom_amd.refblock import coadd_output_stamps, stamp_config
    from pyimcom_amd.select import InStampPool
    from pyimcom_amd.stamps import BlockTables

    n1P, E = 84, 6
    # (PSF grid of 24 native pixels instead of 48: the host evaluates nsamp^2 sampling positions per group and exposure,
    # 11 094 times for this block, as the reference does -- the point here is the block, not the PSF postage stamp)
    wl = dataclasses.replace(synth.CONFIGS["cfg2"], n2=32, fade=3, dtheta_as=0.0390625, n_expo=E, n_inframe=2, npixpsf=24, psf="gauss")  # (Gaussian PSFs: an Airy pattern cut at +-12 pixels makes A indefinite beyond kappa -- every stamp would take the repair path)
    blk, psfgrp, inst, image_at = synth.duck_block(wl, n1P, E, seed=3)
    cfg, ns, nst = blk.cfg, wl.nsamp, n1P + 2
    torch.cuda.empty_cache()
    maps = coadd_output_stamps(blk, psfgrp, flat_penalty=wl.flat_penalty, positions="exact")  # (bit for bit against a provider that evaluates every position)
    assert np.isfinite(blk.out_map).all() and np.abs(blk.out_map).max() > 0 and blk.out_map.shape == (1, 2, maps.nside, maps.nside)
    assert (blk.T_weightmap != 0).all() and blk.kappa_map.min() > 0
    del maps
    torch.cuda.empty_cache()

    # the same block through coadd_block with a provider of its own
    scfg = stamp_config(cfg, psfgrp, E, wl.flat_penalty)
    pool = InStampPool(inst, scfg.n_inframe)
    lin_s = np.arange(ns) - (ns - 1) / 2.0
    gx, gy = np.meshgrid(lin_s, lin_s)
    xy = np.stack([gx.ravel(), gy.ravel()], axis=1) * wl.dscale

    def provider(keys):
        imgs, yx = [], []
        for gj, gi in keys:
            p0 = np.array([2 * gi * wl.n2 - 0.5, 2 * gj * wl.n2 - 0.5])
            for e in range(E):
                imgs.append(image_at(e, p0))
                d = (blk.inimages[e].outpix2world2inpix(xy + p0) - blk.inimages[e].outpix2world2inpix(p0[None])) * wl.oversamp
                yx.append(np.stack([d[:, 1].reshape(ns, ns), d[:, 0].reshape(ns, ns)]))
        return psfs.sample_psf(torch.as_tensor(np.stack(imgs), device="cuda:0"), ns, torch.as_tensor(np.stack(yx), device="cuda:0"), False, True)

    ng = nst // 2
    timg = psfs.get_outpsf("GAUSSIAN", wl.extrasmooth, 2, ns, wl.oversamp, device="cuda:0")
    target = psfs.sample_psf(timg[None], ns, None, False, True)
    tabs = BlockTables({(gj, gi): None for gj in range(ng) for gi in range(ng)}, target, wl.nfft, group_count={(gj, gi): E for gj in range(ng) for gi in range(ng)},
                       bulk_provider=provider, cells=True)
    direct = coadd_block(scfg, pool, tabs, n1P, E, pad_sides="all")
    assert len(plan_block(scfg, pool, tabs, n1P)) >= 28
    assert np.array_equal(direct.out_map.cpu().numpy(), blk.out_map) and np.array_equal(direct.T_weightmap.cpu().numpy(), blk.T_weightmap)
    for k, name in (("UC", "UC_map"), ("Sigma", "Sigma_map"), ("kappa", "kappa_map"), ("Tsum", "Tsum_map"), ("Neff", "Neff_map")):
        assert np.array_equal(direct.maps[k].cpu().numpy(), getattr(blk, name)), k


@pytest.mark.gpu
def test_block_seam_honours_the_stamp_window_nrun_and_outmaps():
    """What the reference's loop coadds and which maps it fills (coadd.py:1808-1842, 1979-1990, 2038-2064): with postage_pad = 1
    and no padded side the window is stamps 2 .. n1P - 1; cfg.stoptile ends the loop after nrun stamps; of the quality maps only
    those named in cfg.outmaps exist on the block.  The adapter follows all three: the windowed call equals an explicit stamp
    list in the reference's cell order; stamps outside the window are left zero; `blk.kappa_map` is not written for "US"."""
    import dataclasses

    from pyimcom_amd import synth
    from pyimcom_amd.blockrun import reference_stamp_order
    from pyimcom_amd.refblock import coadd_output_stamps

    n1P, E = 6, 3
    wl = dataclasses.replace(synth.CONFIGS["small"], n_expo=E)
    blk, psfgrp, _, _ = synth.duck_block(wl, n1P, E, seed=8, pad_sides="")
    blk.cfg.postage_pad = 1
    blk.j_st_min = blk.i_st_min = 2
    blk.j_st_max = blk.i_st_max = n1P - 1
    blk.nrun = 10  # the loop stops inside the third cell (cfg.stoptile)
    blk.cfg.outmaps = "US"
    coadd_output_stamps(blk, psfgrp, flat_penalty=wl.flat_penalty)
    assert hasattr(blk, "UC_map") and hasattr(blk, "Sigma_map") and not hasattr(blk, "kappa_map") and not hasattr(blk, "Tsum_map") and not hasattr(blk, "Neff_map")
    order = reference_stamp_order(2, n1P - 1, 2, n1P - 1, 10)
    assert order[:6] == [(2, 2), (2, 3), (3, 2), (3, 3), (2, 4), (2, 5)] and len(order) == 10
    done = np.zeros((n1P, n1P), bool)
    for j, i in order:
        done[j - 1, i - 1] = True
    assert np.array_equal(blk.T_weightmap[0, 0] != 0, done)  # exactly the visited stamps were coadded
    ref, psfgrp2, _, _ = synth.duck_block(wl, n1P, E, seed=8, pad_sides="")
    ref.cfg.postage_pad = 1
    coadd_output_stamps(ref, psfgrp2, flat_penalty=wl.flat_penalty, stamps=order)
    # same stamps, same origin of the cells?  the explicit list runs with the default origin (1, 1): equal where no more than two
    # stamps overlap; the windowed call is the reference's order.  Compare away from the corners of the stamps' overlaps:
    assert np.allclose(blk.out_map, ref.out_map, rtol=0, atol=4e-7 * np.abs(ref.out_map).max())
    assert np.allclose(blk.UC_map, ref.UC_map, rtol=1e-6, atol=0) and np.allclose(blk.Sigma_map, ref.Sigma_map, rtol=1e-6, atol=0)
    assert np.array_equal(blk.T_weightmap, ref.T_weightmap)
    blk.j_st_max = n1P  # 2 .. 6: five rows
    with pytest.raises(ValueError, match="Size must be even"):
        coadd_output_stamps(blk, psfgrp, flat_penalty=wl.flat_penalty)


@pytest.mark.gpu
def test_sampling_positions_from_a_lattice():
    """the host evaluates the WCS chain on a 17 x 17 lattice per PSF group and exposure instead of at all
    nsamp^2 = 146 689 sampling positions (psfutil.py:751-771); the device forms the positions (imcom_lattice_positions).  On affine maps
    and on maps with quadratic + cubic distortion the positions agree with the direct evaluation to 1e-10 samples (host and device
    memory), and the Block seam's maps with ``positions="lattice"`` (the default) agree with ``positions="exact"`` to the level the
    overlap tables are good to."""
    import torch

    from pyimcom_amd import psfs, synth
    from pyimcom_amd.refblock import LATTICE, coadd_output_stamps

    wl = synth.CONFIGS["cfg2"]
    ns = wl.nsamp
    lin = (np.arange(ns) - (ns - 1) / 2.0) * wl.dscale
    nodes, W = psfs.lattice_nodes_and_weights(lin, LATTICE)
    assert W.shape == (ns, LATTICE) and np.abs(W.sum(axis=1) - 1).max() < 1e-14
    worst = 0.0
    for distortion in (0.0, 1e-7, 1e-5):
        blk, psfgrp, _, _ = synth.duck_block(wl, 4, 3, seed=3, distortion=distortion)
        p0 = np.array([3 * wl.n2 - 0.5, 1 * wl.n2 - 0.5])
        gx, gy = np.meshgrid(lin, lin)
        lx, ly = np.meshgrid(nodes, nodes)
        exact, lat = [], []
        for im in blk.inimages:
            f = lambda pts: (im.outpix2world2inpix(pts + p0) - im.outpix2world2inpix(p0[None])) * wl.oversamp  # noqa: E731
            d = f(np.stack([gx.ravel(), gy.ravel()], axis=1))
            exact.append(np.stack([d[:, 1].reshape(ns, ns), d[:, 0].reshape(ns, ns)]))
            d = f(np.stack([lx.ravel(), ly.ravel()], axis=1))
            lat.append(np.stack([d[:, 1].reshape(LATTICE, LATTICE), d[:, 0].reshape(LATTICE, LATTICE)]))
        exact, lat = np.stack(exact), np.stack(lat)
        got_h = psfs.lattice_positions(lat, W, ns)
        got_d = psfs.lattice_positions(torch.as_tensor(lat, device="cuda:0"), W, ns).cpu().numpy()
        assert np.array_equal(got_h, got_d)
        worst = max(worst, float(np.abs(got_h - exact).max()))
        assert np.abs(got_h - exact).max() < 1e-10, (distortion, np.abs(got_h - exact).max())  # in samples (1/8 native pixel), of offsets up to 200
        if distortion == 1e-5:  # (the distorted map really is not affine over the window: 0.02 samples)
            mid = exact[:, :, ns // 2, :]
            bend = np.abs(mid - (mid[:, :, :1] + (mid[:, :, -1:] - mid[:, :, :1]) * np.linspace(0, 1, ns)[None, None, :])).max()
            assert bend > 1e-3, bend
    # the seam: lattice (default) against exact positions on a distorted block
    wl_s = synth.CONFIGS["small"]
    outs = {}
    for mode in ("lattice", "exact"):
        blk, psfgrp, _, _ = synth.duck_block(wl_s, 4, wl_s.n_expo, seed=5, distortion=1e-6)
        coadd_output_stamps(blk, psfgrp, flat_penalty=wl_s.flat_penalty, positions=mode)
        outs[mode] = (blk.out_map.copy(), blk.UC_map.copy(), blk.Sigma_map.copy(), blk.T_weightmap.copy())
    for a, b in zip(outs["lattice"], outs["exact"]):
        assert np.abs(a - b).max() <= 1e-6 * np.abs(b).max() + 1e-9, np.abs(a - b).max()  # float32 maps: a few ulp
    assert np.abs(outs["exact"][0]).max() > 0
