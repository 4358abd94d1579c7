#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference.

Runs only in the build container (needs /root/reference); nothing of the
reference travels: the outputs are small .npz files holding inputs and the
reference's outputs.  The reference hot-path modules are loaded *by file path*
 with four inert stand-ins for packages the image lacks
(numba.njit = identity decorator, galsim = empty, astropy.units with
degree->arcsec, pyimcom.config.Settings carrying the constants of
src/pyimcom/config.py:85-98).  None of the stand-ins takes part in the
arithmetic that is recorded.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Functions recorded (reference file:line):
  routine.iD5512C_getw 29-122, iD5512C 125-181, iD5512C_sym 184-253,
  gridD5512C 256-338, lakernel1 341-430, lsolve_sps 433-484,
  build_reduced_T_wrap 487-588
  lakernel.CholKernel 226-394 (single/multi kappa, repair path 262-279),
  lakernel.EigenKernel 141-223 (single/multi kappa)
  psfutil.PSFGrp.accel_pad_and_rfft2 943-986, PSFOvl._build_psfovl 1244-1294,
  PSFOvl._call_ii_self 1597-1732, _call_ii_cross 1401-1495,
  _call_io_cross 1497-1595, OutPSF.psf_gaussian 117-146,
  OutPSF.psf_simple_airy 148-223
"""

import importlib.util
import os
import sys
import types
import warnings

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference/src/pyimcom"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load_reference():
    nb = types.ModuleType("numba")
    nb.njit = lambda f=None, **k: f if f is not None else (lambda g: g)
    sys.modules["numba"] = nb
    sys.modules["galsim"] = types.ModuleType("galsim")

    astropy = types.ModuleType("astropy")
    units = types.ModuleType("astropy.units")

    class _Unit:
        def __init__(self, rad):
            self.rad = rad

        def to(self, what):
            tab = {"rad": 1.0, "arcsec": np.pi / 180.0 / 3600.0}
            return self.rad / tab[what]

    units.degree = _Unit(np.pi / 180.0)
    units.arcmin = _Unit(np.pi / 180.0 / 60.0)
    units.arcsec = _Unit(np.pi / 180.0 / 3600.0)
    astropy.units = units
    sys.modules["astropy"] = astropy
    sys.modules["astropy.units"] = units

    pkg = types.ModuleType("pyimcom")
    pkg.__path__ = [REF]
    sys.modules["pyimcom"] = pkg

    cfg = types.ModuleType("pyimcom.config")

    class Settings:  # constants of config.py:85-98
        degree = np.pi / 180.0
        arcmin = degree / 60.0
        arcsec = arcmin / 60.0
        QFilterNative = [1.155, 1.456, 1.250, 1.021, 0.834, 0.689, 0.491, 1.009, 0.000, 1.159, 1.685]
        obsc = 0.31
        pixscale_native = 0.11 * arcsec

    cfg.Settings = Settings
    cfg.format_axis = lambda *a, **k: None
    cfg.format_axis_pars = {}
    sys.modules["pyimcom.config"] = cfg

    mods = {}
    for name in ("routine", "lakernel", "psfutil"):
        spec = importlib.util.spec_from_file_location(f"pyimcom.{name}", f"{REF}/{name}.py")
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"pyimcom.{name}"] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods["routine"], mods["lakernel"], mods["psfutil"]


class Empty:
    pass


def cosine_system(N=6, m=16):
    """The analytic system of tests/pyimcom/test_la.py:49-63."""
    A = np.zeros((N, N))
    d = np.zeros((N, N))
    for i in range(N):
        for j in range(N):
            d[i, j] = 2 * np.pi * (i - j) / N
    for k in range(1, N // 2 + 1):
        A += np.cos(k * d) / k / N
    mBhalf = np.zeros((1, m, N))
    for i in range(N):
        for j in range(m):
            _d = 2 * np.pi * (i - 0.4 * j) / N
            for k in range(1, N // 2 + 1):
                mBhalf[0, j, i] += np.cos(k * _d) / k / N
    return A, mBhalf, A[0, 0]


def gaussian_system(n1, m1, sigma=4.0, off=5.0, step=0.25, scale=0.7):
    """The Gaussian-PSF system of tests/pyimcom/test_routine.py:73-104."""
    n, m = n1 * n1, m1 * m1
    x = np.zeros((n,))
    y = np.zeros((n,))
    for i in range(n1):
        y[n1 * i : n1 * i + n1] = i
        x[i::n1] = i
    xout = np.zeros((m,))
    yout = np.zeros((m,))
    for i in range(m1):
        yout[m1 * i : m1 * i + m1] = off + step * i
        xout[i::m1] = off + step * i
    A = np.exp(-1.0 / sigma**2 * ((x[:, None] - x[None, :]) ** 2 + (y[:, None] - y[None, :]) ** 2))
    mBhalf = np.exp(-1.0 / sigma**2 * ((x[None, :] - xout[:, None]) ** 2 + (y[None, :] - yout[:, None]) ** 2))
    return A * scale, mBhalf * scale, 1.0 * scale


def make_outst(A, mBhalf, C, n2f, kappaC, uctarget, sigmamax):
    cfg = Empty()
    cfg.n2f = n2f
    cfg.n_out = mBhalf.shape[0]
    cfg.kappaC_arr = kappaC
    cfg.uctarget = uctarget
    cfg.sigmamax = sigmamax
    blk = Empty()
    blk.cfg = cfg
    outst = Empty()
    outst.blk = blk
    outst.inpix_cumsum = np.array([A.shape[0]])
    outst.sysmata = A
    outst.mhalfb = mBhalf
    outst.outovlc = np.atleast_1d(np.asarray(C, dtype=np.float64))
    return outst


def run_kernel(K, A, mBhalf, C, n2f, kappaC, uctarget, sigmamax):
    outst = make_outst(A.copy(), mBhalf.copy(), C, n2f, kappaC, uctarget, sigmamax)
    k = K(outst)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        k()
    return dict(T=outst.T, UC=outst.UC, Sigma=outst.Sigma, kappa=outst.kappa)


def main():
    routine, lakernel, psfutil = _load_reference()
    out = {}

    # ---------------------------------------------------------------- IN-1
    fh = np.concatenate([np.linspace(-0.5, 0.5, 41), np.array([0.123456789, -0.3141592653, 0.4999999])])
    w = np.zeros((fh.size, 10))
    for i, f in enumerate(fh):
        routine.iD5512C_getw(w[i], f)
    np.savez_compressed(f"{HERE}/getw.npz", fh=fh, w=w)

    # ---------------------------------------------------------- IN-2/3/4
    # inputs of tests/pyimcom/test_routine.py:11-63 (includes off-grid points)
    nx, ny, N = 32, 64, 10
    npts = N * N
    infunc = np.sin(np.linspace(0, 200, 2 * nx * ny)).reshape((2, ny, nx))
    x_, _ = np.modf(np.arange(npts) / np.sqrt(5))
    x_ *= 40
    y_, _ = np.modf(np.arange(npts) * 2 / np.sqrt(5))
    y_ *= 40
    f1 = np.full((2, npts), -7.0)  # sentinel: off-grid points must stay untouched
    routine.iD5512C(infunc, x_, y_, f1)
    xs, ys = x_.copy(), y_.copy()
    for i in range(1, N):
        for j in range(i):
            xs[i * N + j] = xs[j * N + i]
            ys[i * N + j] = ys[j * N + i]
    f2 = np.full((2, npts), -7.0)
    routine.iD5512C_sym(infunc, xs, ys, f2)
    npi, nxo, nyo = 3, 12, 20
    xpos = np.zeros((npi, nxo))
    ypos = np.zeros((npi, nyo))
    for i in range(npi):
        xpos[i, :] = np.linspace(2 + i, nx - 2 - i, nxo)
        ypos[i, :] = np.linspace(2 + i, ny - 2 - i, nyo)
    f3 = np.full((npi, nxo * nyo), -7.0)
    routine.gridD5512C(infunc[0], xpos, ypos, f3)
    # a second scattered case with random points, all on-grid, 3 layers
    rng = np.random.default_rng(20260723)
    infunc_b = rng.standard_normal((3, 40, 37))
    xb = rng.uniform(4.0, 37 - 5.001, 257)
    yb = rng.uniform(4.0, 40 - 5.001, 257)
    f4 = np.zeros((3, 257))
    routine.iD5512C(infunc_b, xb, yb, f4)
    np.savez_compressed(
        f"{HERE}/interp.npz", infunc=infunc, x=x_, y=y_, f_scatter=f1, xs=xs, ys=ys, f_sym=f2,
        xpos=xpos, ypos=ypos, f_grid=f3, infunc_b=infunc_b, xb=xb, yb=yb, f_scatter_b=f4,
    )

    # ---------------------------------------------------------------- EI-2
    # reduced-size twin of test_routine.py:66-145 (n1=13, m1=9) and the full-size one (33 -> 25)
    for tag, n1, m1, off in (("small", 13, 9, 4.0), ("full", 33, 25, 5.0)):
        A, mB, C = gaussian_system(n1, m1, off=off)
        lam, Q = np.linalg.eigh(A)
        mP = mB @ Q
        m, n = mP.shape
        kap, Sig, UC, T = np.zeros(m), np.zeros(m), np.zeros(m), np.zeros((m, n))
        routine.lakernel1(lam, Q, mP, C, 1e-8, 1e-16, 1e16, 53, kap, Sig, UC, T, 0.5)
        rows = np.array([0, m // 3, m - 1])
        np.savez_compressed(
            f"{HERE}/lakernel1_{tag}.npz", n1=n1, m1=m1, off=off, lam=lam, mPhalf_rows=mP[rows], rows=rows,
            kappa=kap, Sigma=Sig, UC=UC, T_rows=T[rows], Tabsmax=np.abs(T).max(),
            **({"mPhalf": mP, "T": T} if tag == "small" else {}),
        )
        print("lakernel1", tag, kap.min(), kap.max(), Sig.min(), Sig.max(), UC.min(), UC.max(), np.abs(T).max())
        if tag == "small":
            A_ = A + np.identity(n)
            b_ = mB[0, :].copy()
            x_sps = np.zeros(n)
            routine.lsolve_sps(n, A_.copy(), x_sps, b_)
            np.savez_compressed(f"{HERE}/lsolve_sps.npz", A=A_, b=b_, x=x_sps)

    # ---------------------------------------------------------------- CH-3 inner
    rng = np.random.default_rng(20260724)
    m, nv = 40, 3
    kappa_nodes = np.array([1e-5, 1e-4, 1e-3])
    A, mB, C = gaussian_system(9, 7, sigma=1.5, off=2.0, step=0.6)
    mB = mB[: m]
    n = A.shape[0]
    from scipy.linalg import cho_factor, cho_solve
    Tpi = np.zeros((nv, m, n))
    for p in range(nv):
        Tpi[p] = cho_solve(cho_factor(A + kappa_nodes[p] * C * np.identity(n)), mB.T).T
    Dp = np.einsum("ai,pai->ap", mB, Tpi)
    Npq = np.einsum("pai,qai->apq", Tpi, Tpi)
    Epq = np.zeros((m, nv, nv))
    for p in range(nv):
        for q in range(p):
            Epq[:, q, p] = Epq[:, p, q] = Dp[:, q] - kappa_nodes[p] * C * Npq[:, p, q]
        Epq[:, p, p] = Dp[:, p] - kappa_nodes[p] * C * Npq[:, p, p]
    ok, oS, oU, ow = np.zeros(m), np.zeros(m), np.zeros(m), np.zeros(m * nv)
    for ucmin, smax, tag in ((1e-6, 0.5, "a"), (1e-3, 0.6, "b"), (1e-12, 10.0, "c")):
        routine.build_reduced_T_wrap(Npq.flatten(), Dp.flatten() / C, Epq.flatten() / C, kappa_nodes, ucmin, smax, ok, oS, oU, ow)
        out[f"brt_{tag}"] = dict(ucmin=ucmin, smax=smax, kappa=ok.copy(), Sigma=oS.copy(), UC=oU.copy(), w=ow.copy())
    np.savez_compressed(
        f"{HERE}/build_reduced_T.npz", Nflat=Npq.flatten(), Dflat=Dp.flatten() / C, Eflat=Epq.flatten() / C,
        kappa_nodes=kappa_nodes, **{f"{t}_{k}": v for t, d in out.items() for k, v in d.items()},
    )

    # ---------------------------------------------------------------- LA kernels
    la = {}
    A, mB, C = cosine_system()
    la["cos_A"], la["cos_mBhalf"], la["cos_C"] = A, mB, C
    for name, K, kC, uct, smax in (
        ("cos_eig1", lakernel.EigenKernel, [1e-2], 1e-4, 0.5),
        ("cos_eigm", lakernel.EigenKernel, [1e-4, 1e-3, 1e-2], 1e-4, 1.0),
        ("cos_chol1", lakernel.CholKernel, np.array([1e-2]), 1e-4, 0.5),
        ("cos_cholm", lakernel.CholKernel, np.array([1e-4, 1e-3, 1e-2]), 1e-4, 1.0),
    ):
        r = run_kernel(K, A, mB, C, 4, kC, uct, smax)
        for k, v in r.items():
            la[f"{name}_{k}"] = v
        print(name, r["UC"].ravel()[:2], r["Sigma"].ravel()[:2], r["kappa"].ravel()[:2])
    # Gaussian system N=169, m=81 (n2f=9), two target PSFs to exercise n_out=2
    A, mB, C = gaussian_system(13, 9, sigma=2.0, off=4.0, step=0.5)
    mB2 = np.stack([mB, 0.9 * mB[::-1]])
    C2 = np.array([C, 0.8 * C])
    la["gau_A"], la["gau_mBhalf"], la["gau_C"] = A, mB2, C2
    for name, K, kC, uct, smax in (
        ("gau_chol1", lakernel.CholKernel, np.array([6e-4]), 1e-6, 0.5),
        ("gau_cholm", lakernel.CholKernel, np.array([1e-5, 1e-4, 1e-3]), 1e-6, 0.5),
        ("gau_eig1", lakernel.EigenKernel, np.array([6e-4]), 1e-6, 0.5),
        ("gau_eigm", lakernel.EigenKernel, np.array([1e-5, 1e-4, 1e-3]), 1e-6, 0.5),
        ("gau_chol0", lakernel.CholKernel, np.array([0.0]), 1e-6, 0.5),
    ):
        try:
            r = run_kernel(K, A, mB2, C2, 9, kC, uct, smax)
        except Exception as e:  # kappa=0 on a singular Gaussian system may fail outright
            print(name, "reference raised", type(e).__name__, e)
            continue
        for k, v in r.items():
            la[f"{name}_{k}"] = v
        print(name, r["UC"].ravel()[:2], r["Sigma"].ravel()[:2], r["kappa"].ravel()[:2])
    # repair path (tests/pyimcom/test_la.py:8-24) and a solve through it
    A6, mB6, C6 = cosine_system()
    Aneg = A6 - 1e-3 * np.identity(6)
    AA = Aneg + 1e-4 * np.identity(6)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        L = lakernel.CholKernel._cholesky_wrapper(AA, np.diag_indices(6), Aneg)
    la["repair_A"], la["repair_kappa_abs"], la["repair_L"], la["repair_AA_after"] = Aneg, 1e-4, L, AA
    r = run_kernel(lakernel.CholKernel, Aneg, mB6, C6, 4, np.array([1e-4 / C6]), 1e-4, 0.5)
    for k, v in r.items():
        la[f"repair_chol1_{k}"] = v
    # N == 0 special case (lakernel.py:110-119)
    r = run_kernel(lakernel.CholKernel, np.zeros((0, 0)), np.zeros((1, 16, 0)), 1.0, 4, np.array([1e-3]), 1e-4, 0.5)
    for k, v in r.items():
        la[f"empty_{k}"] = v
    np.savez_compressed(f"{HERE}/lakernel.npz", **la)

    # ---------------------------------------------------------------- PSF overlap + A/B sub-blocks
    PSFGrp, PSFOvl, OutPSF = psfutil.PSFGrp, psfutil.PSFOvl, psfutil.OutPSF
    npixpsf, oversamp, dtheta_as = 8, 4, 0.04
    PSFGrp.setup(npixpsf=npixpsf, oversamp=oversamp, dtheta=dtheta_as / 3600.0, psfsplit=False)
    PSFOvl.setup(flat_penalty=1e-7)
    nsamp, nc, nfft = PSFGrp.nsamp, PSFGrp.nc, PSFGrp.nfft
    ps = dict(npixpsf=npixpsf, oversamp=oversamp, dtheta_as=dtheta_as, nsamp=nsamp, nc=nc, nfft=nfft,
              dscale=PSFGrp.dscale, flat_penalty=PSFOvl.flat_penalty)

    def grp_from(psf_arr, in_or_out):
        g = PSFGrp.__new__(PSFGrp)
        g.in_or_out = in_or_out
        g.n_psf = psf_arr.shape[0]
        g.psf_rft = PSFGrp.accel_pad_and_rfft2(psf_arr)
        return g

    yy, xx = PSFGrp.yxo
    def gpsf(sx, sy, th, dx=0.0, dy=0.0):
        c, s = np.cos(th), np.sin(th)
        u = (xx - dx) * c + (yy - dy) * s
        v = -(xx - dx) * s + (yy - dy) * c
        p = np.exp(-0.5 * ((u / sx) ** 2 + (v / sy) ** 2))
        return p / p.sum()

    psf1 = np.stack([gpsf(3.0, 3.3, 0.2), gpsf(3.4, 3.1, 1.0, 0.3, -0.2), gpsf(3.2, 3.2, 0.0, -0.4, 0.1)])
    psf2 = np.stack([gpsf(3.1, 3.5, 0.5, 0.1, 0.1), gpsf(3.3, 3.0, 2.0), gpsf(2.9, 3.2, 0.7, 0.2, 0.3)])
    psfo = np.stack([gpsf(4.0, 4.0, 0.0)])
    g1, g2, go = grp_from(psf1, True), grp_from(psf2, True), grp_from(psfo, False)
    for g in (g1, g2):
        g.idx_blk2grp = np.arange(3, dtype=np.uint8)
        g.idx_grp2blk = np.arange(3, dtype=np.uint8)
    ps["psf1"], ps["psf2"], ps["psfo"] = psf1, psf2, psfo
    ps["rft1"] = g1.psf_rft
    o_self, o_cross = PSFOvl(g1), PSFOvl(g1, g2)
    o_io = PSFOvl(g1, go)
    o_out = PSFOvl(go)
    ps["ovl_self"], ps["ovl_cross"], ps["ovl_io"], ps["outovlc"] = o_self.ovl_arr, o_cross.ovl_arr, o_io.ovl_arr, o_out.outovlc

    # duck-typed InStamps: 3 exposures; counts (4,0,5) and (3,4,2) to exercise empty segments
    rng = np.random.default_rng(20260725)
    blk = Empty()
    blk.n_inimage = 3

    def instamp(counts, x0, y0, j_st, i_st):
        st = Empty()
        st.blk = blk
        st.j_st, st.i_st = j_st, i_st
        st.pix_count = np.array(counts, dtype=np.uint32)
        st.pix_cumsum = np.cumsum([0] + list(counts), dtype=np.uint32)
        nn = int(st.pix_cumsum[-1])
        st.x_val = x0 + rng.uniform(0.0, 6.0, nn)
        st.y_val = y0 + rng.uniform(0.0, 6.0, nn)
        return st

    st1 = instamp([4, 0, 5], 0.0, 0.0, 1, 1)
    st2 = instamp([3, 4, 2], 5.0, 1.0, 1, 2)
    ps["st1_x"], ps["st1_y"], ps["st1_count"] = st1.x_val, st1.y_val, st1.pix_count
    ps["st2_x"], ps["st2_y"], ps["st2_count"] = st2.x_val, st2.y_val, st2.pix_count
    ps["A_self_11"] = o_self(st1, None)
    ps["A_self_12"] = o_self(st1, st2)
    ps["A_cross_12"] = o_cross(st1, st2)
    # output stamp: 6x6 grid of integer output pixels; selection on st1 (subset) and none on st2
    ost = Empty()
    ost.j_st, ost.i_st = 1, 1
    ost.yx_val = np.mgrid[0:6, 2:8].astype(np.float64)
    sel = np.array([0, 2, 3, 5, 8], dtype=np.uint32)
    ost.selections = [None] * 9
    ost.selections[(st1.j_st - ost.j_st + 1) * 3 + (st1.i_st - ost.i_st + 1)] = sel
    go.blk = Empty()
    ps["out_yx"], ps["sel1"] = ost.yx_val, sel
    ps["B_io_1sel"] = o_io(st1, ost)
    ost.selections = [None] * 9
    ps["B_io_1all"] = o_io(st1, ost)

    # target PSFs used by the synthetic configurations (psfutil.py:117-223)
    ps["gauss_16_25"] = OutPSF.psf_gaussian(16, 2.5, 2.5)
    ps["airy_24"] = OutPSF.psf_simple_airy(24, 1.25 * 4, obsc=0.31, tophat_conv=4.0, sigma=0.3 * 4)
    np.savez_compressed(f"{HERE}/psfovl.npz", **ps)
    print("A_self_11 sym err", np.abs(ps["A_self_11"] - ps["A_self_11"].T).max(), "C", ps["outovlc"])

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(f"{HERE}/{f}"))


if __name__ == "__main__":
    main()
