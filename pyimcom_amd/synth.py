"""Synthetic postage-stamp workloads: the inputs of the stamp seam without
FITS/WCS machinery.  Host-side NumPy set-up code: it produces what pyimcom's Block/InStamp layers would
hand to the stamp driver (pixel positions in output-pixel units, pixel values, sampled PSFs); none of it
is on the timed path.

Geometry follows the reference: OutStamp (j_st, i_st) = (1, 1) covers output pixels 0..n2-1 plus
`fade` transition pixels each side (coadd.py:869-882); the nine neighbouring InStamps are the n2 x n2
cells around it (coadd.py:207, 348-349, 853); selection = whole centre cell, edge strips within rho,
corner quarter discs of radius rho = INPAD/dtheta output pixels (coadd.py:716-749, 923-931).
"""

from dataclasses import dataclass, field

import numpy as np

NATIVE_ARCSEC = 0.11  # Settings.pixscale_native (config.py)


@dataclass
class WorkloadConfig:
    name: str
    n2: int
    fade: int
    dtheta_as: float      # output pixel scale, arcsec
    n_expo: int           # exposures (E); a (lo, hi) tuple draws a uniform integer per stamp
    inpad_as: float       # INPAD, arcsec
    kernel: str           # "Cholesky" | "Eigen"
    kappaC: tuple
    psf: str = "airy"     # "gauss" | "airy"
    npixpsf: int = 48
    oversamp: int = 8
    uctarget: float = 1e-6
    sigmamax: float = 0.5
    flat_penalty: float = 1e-7
    n_inframe: int = 2
    extrasmooth: float = 0.934  # target Gaussian sigma in native pixels
    mask_frac: float = 1e-3
    psf_sigma: float = 0.9      # Gaussian input PSF width, native pixels (cfg-1: 0.9 + 0.05 e)
    n_out: int = 1              # target PSFs (OUTPSF + cfg.outpsf_extra); target k is a Gaussian of width extrasmooth * (1 + k/4)
    no_qlt_ctrl: bool = False   # EMPIRNQC (config.py:573): the Empirical kernel without its quality maps -- no A, no B (coadd.py:1020-1025)
    iter_rtol: float = 1.5e-3   # ITERRTOL / ITERMAX (config.py; lakernel.py:397-442), the Iterative kernel's stopping rule
    iter_max: int = 30

    @property
    def n2f(self):
        return self.n2 + 2 * self.fade

    @property
    def m(self):
        return self.n2f**2

    @property
    def nsamp(self):
        return self.npixpsf * self.oversamp - 1

    @property
    def nc(self):
        return self.nsamp // 2

    @property
    def nfft(self):
        return self.npixpsf * self.oversamp * 2

    @property
    def dscale(self):
        arcsec = np.pi / 180.0 / 60.0 / 60.0  # psfutil.py:610 with config.py:85-98
        return ((NATIVE_ARCSEC * arcsec) / arcsec) / self.oversamp / ((self.dtheta_as / 3600.0) * 3600)

    @property
    def rho(self):
        return self.inpad_as / self.dtheta_as  # coadd.py:923


# BASELINE.json configs
CONFIGS = {
    "cfg1": WorkloadConfig("cfg1", 32, 0, 0.0390625, 4, 0.6, "Cholesky", (6e-4,), psf="gauss"),
    "cfg2": WorkloadConfig("cfg2", 48, 0, 1.25 / 48, 6, 0.45, "Cholesky", (6e-4,)),
    "cfg2f": WorkloadConfig("cfg2f", 48, 3, 1.25 / 48, 6, 0.45, "Cholesky", (6e-4,)),
    "cfg3": WorkloadConfig("cfg3", 48, 0, 1.25 / 48, 8, 0.45, "Eigen", (1e-5, 1e-4, 1e-3)),
    "cfg4": WorkloadConfig("cfg4", 48, 0, 1.25 / 48, (6, 10), 0.45, "Cholesky", (6e-4,)),
    "cfg5": WorkloadConfig("cfg5", 48, 0, 1.25 / 48, 16, 0.45, "Cholesky", (6e-4,)),
    # the reference's own benchmark shape (configs/paper4_configs/H158_Chol_benchmark.json: OUTSIZE [80, 32, 0.0390625], FADE 3, PAD 2 ->
    # n1P = 84, INPAD 1.24 -> rho = 31.7 output pixels against n2 = 32 (the guard of coadd.py:1915), NPIXPSF 48, oversamp 8, GAUSSIAN target
    # with that EXTRASMOOTH, KAPPAC [6e-4], FLATPEN 0, five EXTRAINPUT layers + the science layer = 6 input layers) at six exposures:
    # N ~ 6.2k input pixels against m = 38^2 = 1444 outputs, i.e. N / m = 4.3 -- the factorisation is 42 % of the matrix flops
    "paper4": WorkloadConfig("paper4", 32, 3, 0.0390625, 6, 1.24, "Cholesky", (6e-4,), n_inframe=6, extrasmooth=0.934253980316821,
                             flat_penalty=0.0),
    # the reference's DEFAULT configuration (configs/default_config.json, and 70 more of its 152 configs): LAKERNEL Iterative, KAPPAC [0.0],
    # OUTSIZE [80, 32, 0.0390625], FADE 0, INPAD 0.6 -> rho = 15.36 output pixels, ITERRTOL 1.5e-3, ITERMAX 30, NPIXPSF 48, GAUSSIAN target
    # with that EXTRASMOOTH, FLATPEN 0, the science layer + five EXTRAINPUT layers, at six exposures: N ~ 2.8k, m = 1024, ~560 input pixels
    # inside an output pixel's acceptance disc; with kappa = 0 no conjugate-gradient recurrence reaches the tolerance in 30 steps
    "iter_default": WorkloadConfig("iter_default", 32, 0, 0.0390625, 6, 0.6, "Iterative", (0.0,), n_inframe=6, extrasmooth=0.8493218002880191,
                                   flat_penalty=0.0),
    # small cases for parity tests (oracle finishes in seconds)
    "tiny": WorkloadConfig("tiny", 8, 1, 0.11 / 2.5, 3, 0.12, "Cholesky", (6e-4,), psf="gauss", npixpsf=8, oversamp=4,
                           psf_sigma=0.45, extrasmooth=0.6),
    "small": WorkloadConfig("small", 12, 2, 0.11 / 3.0, 4, 0.2, "Cholesky", (6e-4,), psf="gauss", npixpsf=12, oversamp=6,
                            psf_sigma=0.5, extrasmooth=0.7),
    "smallm": WorkloadConfig("smallm", 12, 0, 0.11 / 3.0, 4, 0.2, "Cholesky", (1e-5, 1e-4, 1e-3), psf="gauss", npixpsf=12,
                             oversamp=6, psf_sigma=0.5, extrasmooth=0.7),
}


def _gauss_psf(cfg, sigma_tab, e1=0.0, e2=0.0):
    ns = cfg.nsamp
    c = (ns - 1) / 2.0
    y, x = np.mgrid[0:ns, 0:ns].astype(np.float64)
    x -= c
    y -= c
    u = x * (1 + e1) + y * e2
    v = y * (1 - e1) + x * e2
    p = np.exp(-0.5 * (u * u + v * v) / sigma_tab**2)
    return p / p.sum()


def _airy_psf(cfg, e1=0.0, e2=0.0):
    """Obscured Airy (obsc 0.31, lambda/D = 1.25 native px, QFilterNative[H158]) convolved with the native
    pixel top-hat and a Gaussian of 0.3 px -- the construction of OutPSF.psf_simple_airy
    (psfutil.py:149-223) on sheared coordinates (per-exposure ellipticity)."""
    from scipy.special import jv

    ns, ov = cfg.nsamp, cfg.oversamp
    ldp, obsc, tophat, sigma = 1.250 * ov, 0.31, 1.0 * ov, 0.3 * ov
    kp = 1 + int(np.ceil(tophat + 6 * sigma))
    npad = ns + 2 * kp
    y, x = np.mgrid[(1 - npad) / 2 : (npad - 1) / 2 : npad * 1j, (1 - npad) / 2 : (npad - 1) / 2 : npad * 1j]
    u = x * (1 + e1) + y * e2
    v = y * (1 - e1) + x * e2
    r = np.sqrt(u * u + v * v) / ldp
    I_ = (np.square(jv(0, np.pi * r) + jv(2, np.pi * r) - obsc**2 * (jv(0, np.pi * r * obsc) + jv(2, np.pi * r * obsc)))
          / (4.0 * ldp**2 * (1 - obsc**2)) * np.pi)
    It = np.fft.rfft2(I_)
    uxa = np.linspace(0, 1 - 1 / npad, npad)
    uxa[-(npad // 2):] -= 1
    ux = np.tile(uxa[None, : npad // 2 + 1], (npad, 1))
    uy = np.tile(uxa[:, None], (1, npad // 2 + 1))
    It *= np.exp(-2.0 * np.pi**2 * (np.square(ux * sigma) + np.square(uy * sigma))) * np.sinc(ux * tophat) * np.sinc(uy * tophat)
    I_ = np.fft.irfft2(It, s=(npad, npad))[kp:-kp, kp:-kp]
    return I_ / I_.sum()


def make_psfs(cfg, n_expo, seed=20260723):
    """Sampled input PSFs [E, nsamp, nsamp] (PSFGrp.psf_arr, psfutil.py:838) and the target PSFs [n_out, nsamp, nsamp]."""
    rng = np.random.default_rng(seed)
    psfs = np.zeros((n_expo, cfg.nsamp, cfg.nsamp))
    for e in range(n_expo):
        if cfg.psf == "gauss":
            psfs[e] = _gauss_psf(cfg, (cfg.psf_sigma + 0.05 * e) * cfg.oversamp)
        else:
            ang = rng.uniform(0, np.pi)
            psfs[e] = _airy_psf(cfg, 0.02 * np.cos(2 * ang), 0.02 * np.sin(2 * ang))
    target = np.stack([_gauss_psf(cfg, cfg.extrasmooth * (1.0 + 0.25 * k) * cfg.oversamp) for k in range(cfg.n_out)])
    return psfs, target


@dataclass
class Stamp:
    """One postage stamp at the stamp seam (what OutStamp._process_input_stamps produces)."""
    x: np.ndarray         # [N] f64 input pixel x, output-pixel units (coadd.py:972)
    y: np.ndarray         # [N] f64
    expo: np.ndarray      # [N] int32 exposure index of each pixel
    seg: np.ndarray       # [N] int32 InStamp segment 0..8 (order of coadd.py:853)
    indata: np.ndarray    # [n_inframe, N] f32 (coadd.py:975)
    out_x0: float         # x of output pixel column 0 (left - fade)
    out_y0: float
    n_expo: int
    inpix_cumsum: np.ndarray = field(default=None)  # [10] (coadd.py:937)

    @property
    def n(self):
        return self.x.size


def make_stamp(cfg, stamp_id, n_expo=None):
    """Input pixels of one stamp: per exposure a rotated, dithered regular lattice of native pixels,
    0.1 % masked, selected and ordered as the reference does (9 InStamp segments, exposure-major inside)."""
    rng = np.random.default_rng(20260723 + stamp_id)
    if n_expo is None:
        n_expo = cfg.n_expo if isinstance(cfg.n_expo, int) else int(rng.integers(cfg.n_expo[0], cfg.n_expo[1] + 1))
    n2, rho, p = cfg.n2, cfg.rho, NATIVE_ARCSEC / cfg.dtheta_as
    assert rho <= n2, "INPAD larger than a stamp (coadd.py:1915)"
    left, bottom = 0, 0
    right, top = n2 - 1, n2 - 1
    per_expo = []
    half = int(np.ceil((1.5 * n2 + 2) * np.sqrt(2) / p)) + 2
    for e in range(n_expo):
        th = np.deg2rad(20.0 * e / n_expo + rng.uniform(-2.0, 2.0))
        dx, dy = rng.uniform(0, p, 2)
        jj, ii = np.mgrid[-half : half + 1, -half : half + 1]
        gx, gy = ii.ravel() * p, jj.ravel() * p
        x = (n2 - 1) / 2.0 + dx + np.cos(th) * gx - np.sin(th) * gy
        y = (n2 - 1) / 2.0 + dy + np.sin(th) * gx + np.cos(th) * gy
        keep = rng.uniform(size=x.size) >= cfg.mask_frac
        per_expo.append((x[keep], y[keep]))
    xs, ys, es, ss = [], [], [], []
    cum = [0]
    for idx, (dj, di) in enumerate((dj, di) for dj in (-1, 0, 1) for di in (-1, 0, 1)):
        # InStamp cell (coadd.py:207, 348-349): x in [(1+di-1)*n2 - 0.5, (1+di)*n2 - 0.5)
        x_lo, y_lo = (di * n2) - 0.5, (dj * n2) - 0.5
        x_piv = [left - 0.5, None, right + 0.5][di + 1]
        y_piv = [bottom - 0.5, None, top + 0.5][dj + 1]
        count = 0
        for e, (x, y) in enumerate(per_expo):
            incell = (x >= x_lo) & (x < x_lo + n2) & (y >= y_lo) & (y < y_lo + n2)
            xc, yc = x[incell], y[incell]
            if x_piv is not None or y_piv is not None:
                d2 = np.zeros(xc.shape)
                if x_piv is not None:
                    d2 += np.square(xc - x_piv)
                if y_piv is not None:
                    d2 += np.square(yc - y_piv)
                sel = d2 < rho**2
                xc, yc = xc[sel], yc[sel]
            xs.append(xc); ys.append(yc)
            es.append(np.full(xc.size, e, np.int32)); ss.append(np.full(xc.size, idx, np.int32))
            count += xc.size
        cum.append(cum[-1] + count)
    x, y = np.concatenate(xs), np.concatenate(ys)
    expo, seg = np.concatenate(es), np.concatenate(ss)
    # pixel values: a unit-flux point source near the stamp centre seen through a Gaussian of the
    # exposure's width (frame 0) and white noise (frame 1)
    sx, sy = (n2 - 1) / 2.0 + rng.uniform(-2, 2), (n2 - 1) / 2.0 + rng.uniform(-2, 2)
    sig = (0.9 + 0.02 * expo) * p
    star = np.exp(-0.5 * ((x - sx) ** 2 + (y - sy) ** 2) / sig**2) / (2 * np.pi * (sig / p) ** 2)
    indata = np.zeros((cfg.n_inframe, x.size), np.float32)
    indata[0] = star + rng.normal(0, 1e-3, x.size)
    for f in range(1, cfg.n_inframe):
        indata[f] = rng.normal(0, 1.0, x.size)
    return Stamp(x, y, expo, seg, indata, float(left - cfg.fade), float(bottom - cfg.fade), n_expo,
                 np.array(cum, dtype=np.uint32))


def algorithmic_flops(cfg, n, nv=None):
    """Per-stamp algorithmic work, fp64 flops, by stage."""
    m = cfg.m
    nv = nv or len(cfg.kappaC)
    out = {
        "build_A": 330.0 * n * (n + 1) / 2,
        "build_B": 220.0 * n * m + 110.0 * n * cfg.n2f,
        "epilogue": 2.0 * n * m * cfg.n_inframe + 3.0 * n * m,
    }
    if cfg.kernel == "Cholesky":
        out["chol_factor"] = nv * n**3 / 3.0
        out["chol_solve"] = nv * 2.0 * n * n * m
        out["finalize"] = 4.0 * n * m * nv
    else:
        out["eigh"] = 9.0 * n**3
        out["eigen_gemm"] = 4.0 * n * n * m
        out["lakernel1"] = 6.0 * 14 * n * m
    out["total"] = sum(out.values())
    return out


def make_instamps(cfg, n1P, n_expo, rng, depth=None):
    """InStamps of the (n1P+2)^2 cells (coadd.py:207, 329-358): per exposure a rotated lattice of native pixels binned
    by cell, exposure-major inside a cell, random data.  ``depth`` [nst, nst] int (optional): only the first depth[j, i]
    exposures cover cell (j, i) (a mosaic whose exposure depth varies, BASELINE configs[3])."""
    nst, n2, p = n1P + 2, cfg.n2, NATIVE_ARCSEC / cfg.dtheta_as
    lo, hi = -n2 - 0.5, (n1P + 1) * n2 - 0.5
    c0 = 0.5 * (lo + hi)
    K = int(np.ceil((hi - lo) / np.sqrt(2.0) / p)) + 2  # the lattice covers the block at any rotation
    xs, ys, cell, cnt = [], [], [], np.zeros((n_expo, nst * nst), np.int64)
    for e in range(n_expo):
        th = np.deg2rad(11.0 * e + 3.0)
        g = np.arange(-K, K + 1) * p
        xx, yy = np.meshgrid(g + rng.uniform(0, p), g + rng.uniform(0, p))
        x = (c0 + np.cos(th) * xx - np.sin(th) * yy).ravel()
        y = (c0 + np.sin(th) * xx + np.cos(th) * yy).ravel()
        ok = (x > lo) & (x < hi) & (y > lo) & (y < hi) & (rng.uniform(size=x.size) > 0.01)
        x, y = x[ok], y[ok]
        c = ((y - lo) // n2).astype(np.int64) * nst + ((x - lo) // n2).astype(np.int64)
        if depth is not None:
            keep = e < np.asarray(depth).reshape(-1)[c]
            x, y, c = x[keep], y[keep], c[keep]
        order = np.argsort(c, kind="stable")  # lattice order inside a cell
        xs.append(x[order]); ys.append(y[order]); cell.append(c[order])
        cnt[e] = np.bincount(c, minlength=nst * nst)
    # cell-major, exposure-major inside a cell
    start = np.concatenate([np.zeros((n_expo, 1), np.int64), np.cumsum(cnt, axis=1)], axis=1)
    out = []
    for k in range(nst * nst):
        px = np.concatenate([xs[e][start[e, k] : start[e, k + 1]] for e in range(n_expo)])
        py = np.concatenate([ys[e][start[e, k] : start[e, k + 1]] for e in range(n_expo)])
        cum = np.concatenate([[0], np.cumsum(cnt[:, k])])
        out.append((px, py, rng.standard_normal((cfg.n_inframe, px.size)).astype(np.float32), cum))
    return out


class _Obj:
    """Plain attribute container (the duck-typed stand-ins below)."""


def group_psfs(psfs, gj, gi):
    """PSFs of the 2 x 2 group of InStamps (gj, gi) of a synthetic block: a smooth modulation of the block's analytic PSFs that
    depends on the group, renormalised (every group has PSFs of its own, as a real block's vary with position)."""
    ns = psfs.shape[-1]
    lin = np.arange(ns, dtype=np.float64) - ns // 2
    mod = 1.0 + 0.02 * np.sin(0.05 * lin * (1 + gi % 3))[None, None, :] + 0.02 * np.cos(0.04 * lin * (1 + gj % 5))[None, :, None]
    p = psfs * mod * (1.0 + 1e-3 * ((7 * gj + 3 * gi) % 11))
    return p / p.sum(axis=(1, 2), keepdims=True)


def duck_block(wl, n1P, E, seed=3, kernel="Cholesky", pad_sides="all", distortion=0.0):
    """A duck-typed ``pyimcom.coadd.Block`` + ``PSFGrp`` class attributes for ``refblock.coadd_output_stamps`` without FITS /
    WCS machinery: synthetic InStamps (``make_instamps``), per exposure an affine output-pixel -> input-pixel map (a small
    rotation) and a PSF image that varies smoothly with the position it is asked for, an identity output WCS.  What the
    reference's Block constructor would leave behind (coadd.py:1560-1937), as far as the stamp loop reads it.
    ``distortion`` > 0 adds quadratic and cubic terms to the maps (an optical distortion: ``distortion`` per output pixel^2, and
    1e-4 of that per pixel^3, different per exposure).  Returns (blk, psfgrp, inst, image_at)."""
    ARCSEC = np.pi / 180.0 / 3600.0
    inst = make_instamps(wl, n1P, E, np.random.default_rng(seed))
    base, _ = make_psfs(wl, E)
    ns, nst = wl.nsamp, n1P + 2
    pad = np.zeros((E, ns + 9, ns + 9))
    pad[:, 4 : 4 + ns, 4 : 4 + ns] = base
    lin = np.arange(ns + 9) - (ns + 8) / 2.0
    psfgrp = _Obj()
    psfgrp.npixpsf, psfgrp.oversamp, psfgrp.nsamp, psfgrp.nfft, psfgrp.dscale = wl.npixpsf, wl.oversamp, ns, wl.nfft, wl.dscale
    cfg = _Obj()
    cfg.n1P, cfg.n2, cfg.fade_kernel, cfg.n2f, cfg.n_inframe = n1P, wl.n2, wl.fade, wl.n2f, wl.n_inframe
    cfg.dtheta, cfg.instamp_pad = wl.dtheta_as / 3600.0, wl.inpad_as * ARCSEC
    cfg.linear_algebra, cfg.no_qlt_ctrl, cfg.kappaC_arr, cfg.uctarget, cfg.sigmamax = kernel, False, np.array(wl.kappaC), wl.uctarget, wl.sigmamax
    cfg.psf_circ, cfg.psf_norm, cfg.amp_penalty = False, True, [0.0, 0.0]
    cfg.n_out, cfg.outpsf, cfg.sigmatarget, cfg.use_filter = 1, "GAUSSIAN", wl.extrasmooth, 2
    cfg.outpsf_extra, cfg.sigmatarget_extra, cfg.postage_pad, cfg.psfsplit, cfg.psf_interp = [], [], 0, None, "D5512"
    cfg.flat_penalty = wl.flat_penalty
    blk = _Obj()
    blk.cfg, blk.n_inimage, blk.pad_sides = cfg, E, pad_sides
    blk.outwcs = _Obj()
    blk.outwcs.all_pix2world = lambda arr, origin: np.asarray(arr, dtype=np.float64)
    scale = wl.dtheta_as / NATIVE_ARCSEC

    def image_at(e, point):  # the PSF of exposure e varies smoothly over the block
        u, v = point[0] / (n1P * wl.n2), point[1] / (n1P * wl.n2)
        return pad[e] * (1.0 + 0.02 * np.sin(0.05 * lin * (1 + u))[None, :] + 0.02 * np.cos(0.04 * lin * (1 + v))[:, None])

    blk.inimages = []
    for e in range(E):
        im = _Obj()
        th = 0.004 * (e - E / 2)
        M = scale * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        im.get_psf_pos = (lambda e_: (lambda point, use_shortrange=True: image_at(e_, point)))(e)
        if distortion:
            def warp(xy, M_=M, q_=distortion * (1.0 + 0.1 * e), c_=1e-4 * distortion):
                xy = np.asarray(xy, dtype=np.float64)
                x, y = xy[:, 0], xy[:, 1]
                lin_ = xy @ M_.T
                return lin_ + scale * np.stack([q_ * (x * x - 0.5 * x * y + 0.3 * y * y) + c_ * x * x * y, q_ * (0.7 * x * y - 0.2 * y * y) + c_ * (y**3 - x * y * y)], axis=1)

            im.outpix2world2inpix = warp
        else:
            im.outpix2world2inpix = (lambda M_: (lambda xy: np.asarray(xy) @ M_.T))(M)
        blk.inimages.append(im)
    blk.instamps = [[None] * nst for _ in range(nst)]
    for j in range(nst):
        for i in range(nst):
            st = _Obj()
            st.x_val, st.y_val, st.data, cum = inst[j * nst + i]
            st.pix_cumsum = np.asarray(cum, dtype=np.uint32)
            st.pix_count = np.diff(np.asarray(cum, dtype=np.int64)).astype(np.uint32)
            if j % 2 == 0 and i % 2 == 0:
                st.psf_compute_point_pix = [i * wl.n2 - 0.5, j * wl.n2 - 0.5]  # coadd.py:710-714
            blk.instamps[j][i] = st
    return blk, psfgrp, inst, image_at
