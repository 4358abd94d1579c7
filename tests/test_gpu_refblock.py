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


@pytest.mark.gpu
def test_block_seam_whole_block_and_other_kernels(golden):
    """All n1P x n1P stamps with the boundary recovery, every LA kernel: finite maps of the reference's shapes, the Cholesky
    block equal to the one assembled stamp by stamp (stamps=...) and recovered afterwards."""
    from pyimcom_amd.refblock import coadd_output_stamps

    g = golden("stamp_chain")
    blk, psfgrp = reference_block(g)
    maps = coadd_output_stamps(blk, psfgrp, flat_penalty=float(g["flat_penalty"]), batch=3)
    full = {k: getattr(blk, k).copy() for k in ("out_map", "UC_map", "Sigma_map", "kappa_map", "Tsum_map", "Neff_map", "T_weightmap")}
    n1P = blk.cfg.n1P
    assert full["out_map"].shape == (1, blk.cfg.n_inframe, maps.nside, maps.nside) and full["T_weightmap"].shape == (1, blk.n_inimage, n1P, n1P)
    assert all(np.isfinite(v).all() for v in full.values()) and np.abs(full["out_map"]).max() > 0
    blk2, _ = reference_block(g)
    coadd_output_stamps(blk2, psfgrp, flat_penalty=float(g["flat_penalty"]), batch=1, stamps=[(j, i) for i in range(1, n1P + 1) for j in range(1, n1P + 1)])
    for k, v in full.items():
        assert np.allclose(getattr(blk2, k), v, rtol=1e-5, atol=1e-6 * np.abs(v).max()), k
    for kernel, kC in (("Eigen", [1e-4, 1e-1]), ("Iterative", [3e-2]), ("Empirical", [2e-3])):
        b, _ = reference_block(g, kernel, kC)
        coadd_output_stamps(b, psfgrp, flat_penalty=float(g["flat_penalty"]), batch=4)
        assert np.isfinite(b.out_map).all() and np.isfinite(b.Sigma_map).all(), kernel
        if kernel == "Iterative":
            assert b.UC_map[b.UC_map != 0].min() > 0 and b.Sigma_map.min() >= 0
