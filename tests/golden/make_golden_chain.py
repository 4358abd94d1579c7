#!/usr/bin/env python3
"""End-to-end golden of ONE output postage stamp through the reference's own code (build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_chain.py   ->  tests/golden/stamp_chain.npz

  coadd.py (functions taken out of its syntax tree, the module cannot be imported: asdf / astropy.io / fitsio):
      InStamp.get_inpsfgrp 751-785, InStamp.make_selection 716-749, OutStamp.__init__ 846-885 (reference counting pass
      included), _process_input_stamps 886-977, _build_system_matrices 1002-1122, trapezoid 1222-1292,
      _perform_coaddition 1294-1363
  psfutil.py (imported by file path as in make_golden.py): PSFGrp (input groups per 2x2 InStamps through
      _build_inpsfgrp / _sample_psf, output group), PSFOvl, SysMatA, SysMatB;   lakernel.py: CholKernel
What is duck-typed: the Block (configuration numbers, timer), the InStamp containers (pixel arrays), the InImages
(get_psf_pos returns a fixed oversampled PSF image per exposure, outpix2world2inpix an affine map) and blk.outwcs
(only printed).  A 2 x 2 block of n2 = 4 stamps, 3 exposures, one PSF group lacking exposure 1; OutStamp (1, 2) touches
all four PSF groups.  With --mid also stamp_chain_mid.npz: the same with n2 = 12, npixpsf = 16 and INPAD 0.3 arcsec (N ~ 250 input pixels).
"""

import ast
import contextlib
import io
import os
import sys
from itertools import combinations

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

REF = "/root/reference/src/pyimcom/coadd.py"


def _strip(fn):
    fn.decorator_list, fn.returns = [], None
    for a in fn.args.args + fn.args.kwonlyargs:
        a.annotation = None
    return fn


def main(tag="", npixpsf=8, n2=4, seed=909, inpad_as=0.12):
    _, lakernel, psfutil = mg._load_reference()
    PSFGrp, PSFOvl, SysMatA, SysMatB = psfutil.PSFGrp, psfutil.PSFOvl, psfutil.SysMatA, psfutil.SysMatB
    tree = ast.parse(open(REF).read(), filename=REF)
    cls = {n.name: n for n in tree.body if isinstance(n, ast.ClassDef)}
    fn = lambda c, name: _strip(next(n for n in cls[c].body if isinstance(n, ast.FunctionDef) and n.name == name))  # noqa: E731

    class Stn:
        arcsec = np.pi / 180.0 / 60.0 / 60.0

    class _Deg:
        @staticmethod
        def to(what):
            assert what == "arcsec"
            return 3600.0

    class u:
        degree = _Deg

    ns = {"np": np, "Stn": Stn, "u": u, "combinations": combinations, "PSFGrp": PSFGrp}
    names = [("InStamp", "get_inpsfgrp"), ("InStamp", "make_selection"), ("OutStamp", "__init__"), ("OutStamp", "_process_input_stamps"),
             ("OutStamp", "_build_system_matrices"), ("OutStamp", "trapezoid"), ("OutStamp", "_perform_coaddition")]
    got = {}
    for c, name in names:
        exec(compile(ast.Module(body=[fn(c, name)], type_ignores=[]), REF, "exec"), ns)
        got[(c, name)] = ns[name]

    class InStamp:
        get_inpsfgrp = got[("InStamp", "get_inpsfgrp")]
        make_selection = got[("InStamp", "make_selection")]

    class OutStamp:
        LAKERNEL = {"Cholesky": lakernel.CholKernel, "Eigen": lakernel.EigenKernel, "Iterative": lakernel.IterKernel,
                    "Empirical": lakernel.EmpirKernel}
        __init__ = got[("OutStamp", "__init__")]
        _process_input_stamps = got[("OutStamp", "_process_input_stamps")]
        _build_system_matrices = got[("OutStamp", "_build_system_matrices")]
        trapezoid = staticmethod(got[("OutStamp", "trapezoid")])
        _perform_coaddition = got[("OutStamp", "_perform_coaddition")]

    ns["OutStamp"], ns["InStamp"] = OutStamp, InStamp

    # ---- geometry
    oversamp, dtheta_as = 4, 0.04
    n1P, fade, n_inimage, n_inframe = 2, 1, 3, 2
    PSFGrp.setup(npixpsf=npixpsf, oversamp=oversamp, dtheta=dtheta_as / 3600.0, psfsplit=False)
    PSFOvl.setup(flat_penalty=1e-7)
    rng = np.random.default_rng(seed)
    out = dict(npixpsf=npixpsf, oversamp=oversamp, dtheta_as=dtheta_as, pars=np.array([n1P, n2, fade, n_inimage, n_inframe]),
               flat_penalty=1e-7, kappaC=np.array([2e-3]), instamp_pad_as=inpad_as)

    cfg = mg.Empty()
    cfg.n1P, cfg.n2, cfg.fade_kernel, cfg.n2f, cfg.n_inframe = n1P, n2, fade, n2 + 2 * fade, n_inframe
    cfg.dtheta, cfg.instamp_pad = dtheta_as / 3600.0, inpad_as * Stn.arcsec
    cfg.linear_algebra, cfg.no_qlt_ctrl, cfg.tempfile = "Cholesky", False, None
    cfg.kappaC_arr, cfg.uctarget, cfg.sigmamax = out["kappaC"], 1e-6, 0.5
    cfg.psf_circ, cfg.psf_norm, cfg.amp_penalty = True, True, [0.0, 0.0]
    cfg.n_out, cfg.outpsf, cfg.sigmatarget, cfg.use_filter = 1, "GAUSSIAN", 1.1, 2
    cfg.outpsf_extra, cfg.sigmatarget_extra = [], []
    blk = mg.Empty()
    blk.cfg, blk.n_inimage, blk.this_sub, blk.timer = cfg, n_inimage, 0, (lambda: 0.0)
    blk.outwcs = mg.Empty()
    blk.outwcs.all_pix2world = lambda arr, origin: np.asarray(arr, dtype=np.float64)

    # InImages: a fixed oversampled PSF image and an affine output-pixel -> input-pixel map per exposure
    blk.inimages = []
    ny, nx = 11 * npixpsf // 2, 5 * npixpsf
    yy, xx = np.mgrid[:ny, :nx]
    for e in range(n_inimage):
        im = mg.Empty()
        im.idsca = (100 + e, 1)
        psf = np.exp(-((xx - 0.48 * nx - 0.3 * e) ** 2 / (26.0 + 3 * e) + (yy - 0.49 * ny + 0.2 * e) ** 2 / (22.0 + 2 * e))) + 0.01 * rng.standard_normal((ny, nx))
        th = 0.15 + 0.4 * e
        M = (dtheta_as / 0.11) * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        t0 = np.array([500.0 + 30 * e, 700.0 - 20 * e])
        im.get_psf_pos = (lambda p: (lambda point, use_shortrange=True: p))(psf)
        im.outpix2world2inpix = (lambda M_, t_: (lambda xy: np.asarray(xy) @ M_.T + t_))(M, t0)
        blk.inimages.append(im)
        out[f"inpsf{e}"], out[f"inM{e}"], out[f"int0{e}"] = psf, M, t0

    # InStamps: lattices of native pixels (pitch 0.11 / 0.04 = 2.75 output pixels) binned in cells of n2; the four
    # InStamps of PSF group (0, 2) hold no pixel of exposure 1
    nst = n1P + 2
    pitch = 0.11 / dtheta_as
    lo = -n2 - 0.5
    blk.instamps = [[None] * nst for _ in range(nst)]
    cells = [[[] for _ in range(nst)] for _ in range(nst)]
    for e in range(n_inimage):
        th = 0.15 + 0.4 * e
        g = np.arange(-14, 15) * pitch
        gx, gy = np.meshgrid(g + rng.uniform(0, pitch), g + rng.uniform(0, pitch))
        x = (np.cos(th) * gx - np.sin(th) * gy).ravel() + n2
        y = (np.sin(th) * gx + np.cos(th) * gy).ravel() + n2
        ci, cj = np.floor((x - lo) / n2).astype(int), np.floor((y - lo) / n2).astype(int)
        for j in range(nst):
            for i in range(nst):
                m = (ci == i) & (cj == j)
                if e == 1 and j < 2 and i >= 2:
                    m[:] = False
                cells[j][i].append((x[m], y[m]))
    for j in range(nst):
        for i in range(nst):
            st = InStamp()
            st.blk, st.j_st, st.i_st = blk, j, i
            parts = cells[j][i]
            st.pix_count = np.array([len(p[0]) for p in parts], dtype=np.uint32)
            st.pix_cumsum = np.cumsum([0] + [len(p[0]) for p in parts], dtype=np.uint32)
            st.x_val, st.y_val = np.hstack([p[0] for p in parts]), np.hstack([p[1] for p in parts])
            st.data = rng.standard_normal((n_inframe, st.x_val.size)).astype(np.float32)
            if j % 2 == 0 and i % 2 == 0:  # coadd.py:710-714
                st.psf_compute_point_pix = [i * n2 - 0.5, j * n2 - 0.5]
                st.inpsfgrp, st.inpsfgrp_ref = None, 0
            blk.instamps[j][i] = st
            out[f"in{j}{i}_x"], out[f"in{j}{i}_y"], out[f"in{j}{i}_data"], out[f"in{j}{i}_cum"] = st.x_val, st.y_val, st.data, st.pix_cumsum

    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        blk.outpsfgrp = PSFGrp(in_or_out=False, blk=blk)
        blk.outpsfovl = PSFOvl(blk.outpsfgrp, None)
        blk.sysmata, blk.sysmatb = SysMatA(blk), SysMatB(blk)
        blk.outstamps = [[None] * nst for _ in range(nst)]
        j_st, i_st = 1, 2
        ost = OutStamp(blk, j_st, i_st)          # reference-counting pass + _process_input_stamps
        blk.outstamps[j_st][i_st] = ost
        blk.sysmata.iisubmats.clear()            # what Block.coadd_output_stamps does after its sim_mode pass (coadd.py:2063-2065)
        blk.sysmatb.iopsfovls.clear()
        ost._build_system_matrices(save_abc=True)
        A, mB, C = ost.sysmata.copy(), ost.mhalfb.copy(), np.array(ost.outovlc, dtype=np.float64)
        UC, Sigma, kappa = ost.UC.copy(), ost.Sigma.copy(), ost.kappa.copy()
        T_raw = ost.T.copy()
        yx_val, iny_val, inx_val = ost.yx_val.copy(), ost.iny_val.copy(), ost.inx_val.copy()
        # the other LA kernels on the same (realistic) A, -B/2, C: lakernel.py 141-223, 325-394, 533-744, 747-805
        cfg.iter_rtol, cfg.iter_max = 1.5e-3, 30
        alt = {}
        for akey, kern, kC in (("eig1", "Eigen", [2e-3]), ("eig2", "Eigen", [1e-4, 1e-1]), ("chol3", "Cholesky", [1e-4, 1e-3, 1e-2]),
                              ("iter1", "Iterative", [3e-2]), ("emp", "Empirical", [2e-3])):
            cfg.linear_algebra, cfg.kappaC_arr = kern, np.array(kC)
            K = OutStamp.LAKERNEL[kern](ost)
            K()
            alt[akey] = (np.array(kC), ost.T.copy(), ost.UC.copy(), ost.Sigma.copy(), ost.kappa.copy())
        cfg.linear_algebra, cfg.kappaC_arr = "Cholesky", out["kappaC"]
        ost.T, ost.UC, ost.Sigma, ost.kappa = T_raw.copy(), UC.copy(), Sigma.copy(), kappa.copy()
        ost._perform_coaddition(save_t=True)
    out.update(j_st=j_st, i_st=i_st, A=A, mBhalf=mB, C=C, UC=UC, Sigma=Sigma, kappa=kappa, T_raw=T_raw, T=ost.T, outimage=ost.outimage,
               Tsum_stamp=ost.Tsum_stamp, Tsum_inpix=ost.Tsum_inpix, Neff=ost.Neff, inpix_cumsum=ost.inpix_cumsum)
    for k, (kC, T_, UC_, Sg_, kp_) in alt.items():
        out[f"{k}_kappaC"], out[f"{k}_T"], out[f"{k}_UC"], out[f"{k}_Sigma"], out[f"{k}_kappa"] = kC, T_, UC_, Sg_, kp_
    out["yx_val"], out["iny_val"], out["inx_val"] = yx_val, iny_val, inx_val
    np.savez_compressed(f"{HERE}/stamp_chain{tag}.npz", **out)
    print("N =", A.shape[0], "cumsum", ost.inpix_cumsum, "C", C, "UC range", float(UC.min()), float(UC.max()))
    print("A sym err", float(np.abs(A - A.T).max()), "lam min", float(np.linalg.eigvalsh(A)[0]))


if __name__ == "__main__":
    main()                                          # N = 29: every branch, seconds
    if "--mid" in sys.argv:                         # N ~ 250, PSF tables wide enough for the whole stamp: minutes (interpreted loops)
        main("_mid", npixpsf=16, n2=12, seed=911, inpad_as=0.3)
