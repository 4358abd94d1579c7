"""The CPU oracle against golden vectors produced by running the reference (tests/golden/make_golden.py)."""

import numpy as np
import pytest

from oracle import oracle as orc
from tests.golden.make_golden import cosine_system, gaussian_system


def test_getw(golden):
    g = golden("getw")
    for fh, w_ref in zip(g["fh"], g["w"]):
        w = np.zeros(10)
        orc.iD5512C_getw(w, float(fh))
        assert np.array_equal(w, w_ref)  # same operation order, no FMA contraction -> bit identical
    w = np.zeros(10)
    orc.iD5512C_getw(w, 0.5)  # tests/pyimcom/test_psf.py:55-61: a delta at tap 5
    assert abs(w[5] - 1) < 1e-8 and np.abs(np.delete(w, 5)).max() < 1e-8


def test_interp(golden):
    g = golden("interp")
    f = np.full_like(g["f_scatter"], -7.0)
    orc.iD5512C(g["infunc"], g["x"], g["y"], f)
    assert np.array_equal(f, g["f_scatter"])
    assert np.abs(f).max() > 0.98 and (f == -7.0).any()  # off-grid points untouched
    f = np.full_like(g["f_sym"], -7.0)
    orc.iD5512C_sym(g["infunc"], g["xs"], g["ys"], f)
    assert np.array_equal(f, g["f_sym"])
    f = np.full_like(g["f_grid"], -7.0)
    orc.gridD5512C(np.ascontiguousarray(g["infunc"][0]), g["xpos"], g["ypos"], f)
    assert np.array_equal(f, g["f_grid"])
    f = np.zeros_like(g["f_scatter_b"])
    orc.iD5512C(g["infunc_b"], g["xb"], g["yb"], f)
    assert np.array_equal(f, g["f_scatter_b"])


@pytest.mark.parametrize("tag", ["small", "full"])
def test_lakernel1(golden, tag):
    g = golden(f"lakernel1_{tag}")
    A, mB, C = gaussian_system(int(g["n1"]), int(g["m1"]), off=float(g["off"]))
    lam, Q = np.linalg.eigh(A)
    mP = np.ascontiguousarray(mB @ Q)
    assert np.allclose(lam, g["lam"], rtol=0, atol=1e-13)
    m, n = mP.shape
    k, S, U, T = np.zeros(m), np.zeros(m), np.zeros(m), np.zeros((m, n))
    # use the golden lam / mPhalf where stored so that eigh's sign freedom cannot matter
    if "mPhalf" in g:
        mP = np.ascontiguousarray(g["mPhalf"])
    orc.lakernel1(np.ascontiguousarray(g["lam"]), Q, mP, C, 1e-8, 1e-16, 1e16, 53, k, S, U, T, 0.5)
    if "mPhalf" in g:
        assert np.array_equal(k, g["kappa"]) and np.array_equal(S, g["Sigma"]) and np.array_equal(U, g["UC"])
        assert np.array_equal(T, g["T"])
    else:  # eigenvectors recomputed here: compare at the reference's own C-vs-numba tolerances
        assert np.abs(k - g["kappa"]).max() < 1e-12 and np.abs(S - g["Sigma"]).max() < 1e-7
        assert np.abs(U - g["UC"]).max() < 1e-13
        # known-answer windows of tests/pyimcom/test_routine.py:131-144
        assert 2.5e-7 < k.min() and k.max() < 3.5e-7
        assert 0.34 < S.min() and S.max() < 0.38
        assert 9e-9 < U.min() and U.max() < 1.1e-8
        assert 0.077 < np.abs(T).max() < 0.079


def test_lsolve_sps(golden):
    g = golden("lsolve_sps")
    x = np.zeros_like(g["x"])
    orc.lsolve_sps(g["A"].shape[0], g["A"].copy(), x, g["b"])
    assert np.array_equal(x, g["x"])
    assert np.abs(x - np.linalg.solve(g["A"], g["b"])).max() < 1e-10


def test_build_reduced_T(golden):
    g = golden("build_reduced_T")
    m = g["brt_a_kappa"].size
    nv = g["kappa_nodes"].size
    for tag in "abc":
        ok, oS, oU, ow = np.zeros(m), np.zeros(m), np.zeros(m), np.zeros(m * nv)
        orc.build_reduced_T_wrap(g["Nflat"], g["Dflat"], g["Eflat"], g["kappa_nodes"], float(g[f"brt_{tag}_ucmin"]),
                                 float(g[f"brt_{tag}_smax"]), ok, oS, oU, ow)
        assert np.array_equal(ok, g[f"brt_{tag}_kappa"]) and np.array_equal(ow, g[f"brt_{tag}_w"])
        assert np.array_equal(oS, g[f"brt_{tag}_Sigma"]) and np.array_equal(oU, g[f"brt_{tag}_UC"])


LA_CASES = [
    ("cos_chol1", "Cholesky", "cos", [1e-2], 1e-4, 0.5, 4), ("cos_cholm", "Cholesky", "cos", [1e-4, 1e-3, 1e-2], 1e-4, 1.0, 4),
    ("cos_eig1", "Eigen", "cos", [1e-2], 1e-4, 0.5, 4), ("cos_eigm", "Eigen", "cos", [1e-4, 1e-3, 1e-2], 1e-4, 1.0, 4),
    ("gau_chol1", "Cholesky", "gau", [6e-4], 1e-6, 0.5, 9), ("gau_cholm", "Cholesky", "gau", [1e-5, 1e-4, 1e-3], 1e-6, 0.5, 9),
    ("gau_eig1", "Eigen", "gau", [6e-4], 1e-6, 0.5, 9), ("gau_eigm", "Eigen", "gau", [1e-5, 1e-4, 1e-3], 1e-6, 0.5, 9),
]


@pytest.mark.parametrize("name,kind,sysn,kC,uct,smax,n2f", LA_CASES)
def test_la_kernels(golden, name, kind, sysn, kC, uct, smax, n2f):
    g = golden("lakernel")
    A, mB, C = g[f"{sysn}_A"], g[f"{sysn}_mBhalf"], np.atleast_1d(g[f"{sysn}_C"])
    T, UC, Sigma, kappa = orc.la_kernel(kind, A, mB, C, n2f, np.array(kC), uct, smax)
    # same LAPACK underneath: agreement to rounding (eigen path: eigenvector sign/degeneracy freedom)
    tolT = 2e-6 * np.abs(g[f"{name}_T"]).max()
    assert np.abs(T - g[f"{name}_T"]).max() <= tolT
    assert np.allclose(UC, g[f"{name}_UC"], rtol=2e-5, atol=1e-9)
    assert np.allclose(Sigma, g[f"{name}_Sigma"], rtol=2e-5, atol=1e-9)
    assert np.allclose(kappa, g[f"{name}_kappa"], rtol=1e-6, atol=0)


def indef_case(golden, name):
    """Inputs of tests/golden/eigen_indef.npz: the existing fixtures' systems minus shift * I (make_golden_eigen_indef.py)."""
    g = golden("eigen_indef")
    if name == "chain":
        ch = golden("stamp_chain_mid")
        A, mB, C, n2f = ch["A"], ch["mBhalf"], ch["C"], 14
    else:
        la = golden("lakernel")
        A, mB, C, n2f = (la["repair_A"], la["cos_mBhalf"], np.atleast_1d(la["cos_C"]), 4) if name == "repair" else (la["gau_A"], la["gau_mBhalf"], la["gau_C"], 9)
    return g, A - float(g[f"{name}_shift"]) * np.identity(A.shape[0]), mB, C, n2f


@pytest.mark.parametrize("name", ["repair", "gau", "chain"])
@pytest.mark.parametrize("tag", ["eig1", "eigm"])
def test_eigen_kernel_on_indefinite_matrices(golden, name, tag):
    """EigenKernel on A with eigenvalues below -kappa (lakernel.py:154-223 divides by lam + kappa whatever its sign): the
    oracle against outputs of the reference itself -- a 6 x 6 known system, a Gaussian one with two target PSFs, and a real
    PSF-overlap matrix (N = 220) with 133 of its eigenvalues negative."""
    g, A, mB, C, n2f = indef_case(golden, name)
    kC = g[f"{name}_{tag}_kappaC"]
    assert np.linalg.eigvalsh(A)[0] + kC[0] * C.min() < 0
    T, UC, Sigma, kappa = orc.la_kernel("Eigen", A, mB, C, n2f, kC, float(g[f"{name}_uctarget"]), float(g[f"{name}_sigmamax"]))
    ref = {k: g[f"{name}_{tag}_{k}"] for k in ("T", "UC", "Sigma", "kappa")}
    assert np.abs(T - ref["T"]).max() <= 2e-6 * np.abs(ref["T"]).max()
    assert np.allclose(UC, ref["UC"], rtol=2e-5, atol=1e-9) and np.allclose(Sigma, ref["Sigma"], rtol=2e-5, atol=1e-9)
    assert np.allclose(kappa, ref["kappa"], rtol=1e-6, atol=0)


def test_la_known_answers(golden):
    """The range assertions of tests/pyimcom/test_la.py:92-99 and 153-160 on the oracle's outputs."""
    A, mB, C = cosine_system()
    T, UC, Sigma, kappa = orc.la_kernel("Eigen", A, mB, np.array([C]), 4, [1e-2], 1e-4, 0.5)
    for j in range(16):
        assert (UC.ravel()[j] < 1e-4) if j % 5 == 0 else (0.05 < UC.ravel()[j] < 0.2)
        assert 0.6 < Sigma.ravel()[j] < 1.0 and 0.002 < kappa.ravel()[j] < 0.004
    T, UC, Sigma, kappa = orc.la_kernel("Eigen", A, mB, np.array([C]), 4, [1e-4, 1e-3, 1e-2], 1e-4, 1.0)
    for j in range(16):
        if j % 5 == 0:
            assert UC.ravel()[j] < 1e-4 and 5e-4 < kappa.ravel()[j] < 1.5e-3
        else:
            assert 0.05 < UC.ravel()[j] < 0.2 and 5e-6 < kappa.ravel()[j] < 1.5e-5


def test_repair_and_empty(golden):
    g = golden("lakernel")
    A = g["repair_A"]
    AA = A + g["repair_kappa_abs"] * np.identity(6)
    L, rep = orc.cholesky_wrapper(AA, A)
    assert rep and np.allclose(L, g["repair_L"], rtol=0, atol=1e-14)
    assert np.allclose(AA, g["repair_AA_after"], rtol=0, atol=0)
    w = np.linalg.eigvalsh(L @ L.T)
    assert abs(w[0] - 1e-4) < 1e-7  # tests/pyimcom/test_la.py:24
    T, UC, Sigma, kappa = orc.la_kernel("Cholesky", np.zeros((0, 0)), np.zeros((1, 16, 0)), np.array([1.0]), 4, [1e-3], 1e-4, 0.5)
    assert T.shape == (1, 16, 0) and np.array_equal(UC, g["empty_UC"]) and np.array_equal(kappa, g["empty_kappa"])
    assert np.array_equal(Sigma, g["empty_Sigma"])


def test_psf_overlap_and_subblocks(golden):
    g = golden("psfovl")
    geo = orc.Geom(int(g["npixpsf"]), int(g["oversamp"]), float(g["dtheta_as"]) / 3600.0, float(g["flat_penalty"]))
    assert geo.nsamp == int(g["nsamp"]) and geo.nc == int(g["nc"]) and geo.nfft == int(g["nfft"])
    assert geo.dscale == float(g["dscale"])
    r1, r2, ro = orc.pad_and_rfft2(g["psf1"], geo), orc.pad_and_rfft2(g["psf2"], geo), orc.pad_and_rfft2(g["psfo"], geo)
    assert np.array_equal(r1, g["rft1"])
    o_self, o_cross, o_io = orc.overlap_self(r1, geo), orc.overlap_cross(r1, r2, geo), orc.overlap_cross(r1, ro, geo)
    assert np.array_equal(o_self, g["ovl_self"]) and np.array_equal(o_cross, g["ovl_cross"])
    assert np.array_equal(o_io, g["ovl_io"]) and np.array_equal(orc.overlap_out_C(ro, geo), g["outovlc"])
    c1, c2 = g["st1_count"], g["st2_count"]
    A11 = orc.subblock_ii_self(o_self, 3, geo, g["st1_x"], g["st1_y"], c1)
    assert np.array_equal(A11, g["A_self_11"]) and np.array_equal(A11, A11.T)
    assert np.array_equal(orc.subblock_ii_self(o_self, 3, geo, g["st1_x"], g["st1_y"], c1, g["st2_x"], g["st2_y"], c2), g["A_self_12"])
    assert np.array_equal(orc.subblock_ii_cross(o_cross, geo, g["st1_x"], g["st1_y"], c1, g["st2_x"], g["st2_y"], c2), g["A_cross_12"])
    ox, oy = g["out_yx"][1, 0, :], g["out_yx"][0, :, 0]
    assert np.array_equal(orc.subblock_io(o_io, geo, g["st1_x"], g["st1_y"], c1, ox, oy), g["B_io_1all"])
    assert np.array_equal(orc.subblock_io(o_io, geo, g["st1_x"], g["st1_y"], c1, ox, oy, g["sel1"].astype(int)), g["B_io_1sel"])
    assert np.array_equal(orc.psf_gaussian(16, 2.5, 2.5), g["gauss_16_25"])
    assert np.allclose(orc.psf_simple_airy(24, 5.0, obsc=0.31, tophat_conv=4.0, sigma=1.2), g["airy_24"], rtol=0, atol=1e-17)


def test_trapezoid_and_coaddition():
    """coadd.py:1222-1354 restated from the text: hand-computed weights and linear-algebra identities."""
    a = np.ones((2, 9, 9))
    orc.trapezoid(a, 2)
    s = np.arange(1, 5) / 5.0
    s = s - np.sin(2 * np.pi * s) / (2 * np.pi)
    assert np.allclose(a[0, 4, :4], s) and np.allclose(a[0, 4, :-5:-1], s) and np.allclose(a[1, :4, 4], s)
    assert np.isclose(a[0, 0, 0], s[0] ** 2) and np.isclose(a[0, 8, 1], s[0] * s[1]) and a[0, 4, 4] == 1.0
    rng = np.random.default_rng(5)
    n2f, n2, N, E = 6, 4, 23, 3
    T = rng.standard_normal((1, n2f * n2f, N)).astype(np.float32)
    indata = rng.standard_normal((2, N)).astype(np.float32)
    expo = np.sort(rng.integers(0, E, N))
    T0 = T.copy()
    out, Ts, Tin, Neff = orc.perform_coaddition(T, indata, expo, E, n2f, n2, 1)
    taper = np.ones((n2f, n2f))
    orc.trapezoid(taper, 1)
    Tt = T0[0] * taper.ravel()[:, None]
    assert np.allclose(out[0, 1].ravel(), Tt @ indata[1], rtol=1e-5, atol=1e-5)
    assert np.allclose(Tin.ravel(), Tt.sum(axis=1), rtol=1e-5, atol=1e-5)
    assert np.allclose(Ts[0], [Tt[:, expo == e].sum() / n2**2 for e in range(E)], rtol=1e-5, atol=1e-5)
    assert (Neff[0, 2:4, 2:4] >= 1 - 1e-9).all() and (Neff[0, 2:4, 2:4] <= E + 1e-9).all()


def test_iter_and_empir_kernels_golden(golden):
    """lakernel.IterKernel 533-744 / EmpirKernel 747-805: the numpy restatement reproduces the reference's outputs
    bit for bit (same BLAS underneath), including the NaN pixels of the approximate multi-kappa U/C."""
    g = golden("lakernel_iter")
    A, mB, C = g["A"], g["mBhalf"], g["C"]
    oy, ox = g["yx"][0].ravel(), g["yx"][1].ravel()
    geo = (oy, ox, g["iny"], g["inx"], float(g["rho_acc"]))
    for name, kC, exact in (("iter1", [6e-4], None), ("iterm", [1e-5, 1e-4, 1e-3], None), ("iter1_exact", [6e-4], True),
                            ("iterm_approx", [1e-5, 1e-4, 1e-3], False)):
        for j in range(2):
            T, UC, S, k, _ = orc.iter_kernel(A, mB[j], C[j], kC, 1e-6, 0.5, *geo, rtol=float(g["rtol"]), maxiter=int(g["itmax"]),
                                             exact_UC=exact)
            assert np.array_equal(T, g[f"{name}_T"][j], equal_nan=True), (name, j)
            for got, key in ((UC, "UC"), (S, "Sigma"), (k, "kappa")):
                assert np.array_equal(got, g[f"{name}_{key}"][j].ravel(), equal_nan=True), (name, j, key)
    for name, nq in (("empir", False), ("empir_noqc", True)):
        for j in range(2):
            T, UC, S, k, _ = orc.empir_kernel(A, mB[j], C[j], [6e-4], *geo, no_qlt_ctrl=nq)
            assert np.array_equal(T, g[f"{name}_T"][j])
            for got, key in ((UC, "UC"), (S, "Sigma"), (k, "kappa")):
                assert np.array_equal(got, g[f"{name}_{key}"][j].ravel()), (name, j, key)


def test_psf_sampling_golden(golden):
    """PSFGrp._sample_psf 709-795 (both interpolator paths), _get_outpsf 854-896 and the cut-out / normalisation
    of PSFGrp.__init__ 650-656: bit-exact against the reference's arrays."""
    g = golden("psf_sample")
    ns, ov = int(g["nsamp"]), int(g["oversamp"])
    assert np.array_equal(orc.sample_psf(g["in_psf"], ns, g["in_yxco"]), g["in_psf_arr"])
    for k in range(2):
        assert np.array_equal(orc.sample_psf(g["grid_psf"][k], ns), g["grid_psf_arr"][k])
    for name, kind in (("gauss", "GAUSSIAN"), ("airyobsc", "AIRYOBSC"), ("airyunobsc", "AIRYUNOBSC")):
        sig, uf, circ, norm = g[f"out_{name}_pars"]
        arr = orc.sample_psf(orc.get_outpsf(kind, sig, int(uf), ns, ov), ns)[None].copy()
        orc.finish_psf_group(arr, bool(circ), bool(norm))
        assert np.array_equal(arr[0], g[f"out_{name}"]), name


def test_smooth_and_pad_golden(golden):
    """InImage.smooth_and_pad (coadd.py:433-474): the restatement against the reference function executed on five
    images (square / rectangular / odd sizes; top-hat only, Gaussian only, both, neither)."""
    g = golden("smooth_pad")
    for name in "abcde":
        w, sg = g[f"{name}_pars"]
        got = orc.smooth_and_pad(g[f"{name}_in"], float(w), float(sg))
        ref = g[f"{name}_out"]
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 4e-16 * np.abs(ref).max(), name


def test_coadd_stamp_functions_golden(golden):
    """The stamp-driver functions of coadd.py -- make_selection / _process_input_stamps (716-749, 886-977), trapezoid
    in both modes (1222-1292), _perform_coaddition (1294-1363), compress_map (2087-2138) -- against outputs of the
    reference's own code (tests/golden/make_golden_coadd.py executes the function definitions taken from coadd.py's
    syntax tree).  Bit for bit, except the einsum / sum reductions of the coaddition (numpy's own reduction order is
    what it is: a few float32 / float64 ulps)."""
    g = golden("coadd_stamp")
    a = g["trap_in"].copy()
    orc.trapezoid(a, 2)
    assert np.array_equal(a, g["trap_out"])
    b = g["recover_in"].copy()
    orc.trapezoid_recover(b, 2, tuple(int(v) for v in g["recover_pads"]))
    assert b.dtype == np.float32 and np.array_equal(b, g["recover_out"])

    bottom, top, left, right = (int(v) for v in g["sel_box"])
    rpix = float(g["sel_rpix"])
    nine, pivots = [], []
    for k, (j, i) in enumerate(g["sel_order"]):
        nine.append((g[f"in{j}{i}_x"], g[f"in{j}{i}_y"], g[f"in{j}{i}_data"], g[f"in{j}{i}_cum"]))
        pivots.append(([left - 0.5, None, right + 0.5][i - 2 + 1], [bottom - 0.5, None, top + 0.5][j - 2 + 1]))  # coadd.py:927-928
        sel = orc.select_pixels(nine[-1][0], nine[-1][1], pivots[-1], rpix)
        ref = g[f"sel_idx{k}"]
        assert (sel is None and ref[0] == -1 and ref.size == 1) or np.array_equal(sel, ref.astype(np.int64)), k
    x, y, indata, expo, cum = orc.process_input_stamps(nine, pivots, rpix)
    assert np.array_equal(cum, g["sel_cumsum"]) and np.array_equal(x, g["sel_x"]) and np.array_equal(y, g["sel_y"])
    assert np.array_equal(indata, g["sel_data"])

    n2, fade, n_expo, n_inframe = (int(v) for v in g["co_pars"])
    T = g["co_T_in"].copy()
    outimage, Tsum_stamp, Tsum_inpix, Neff = orc.perform_coaddition(T, indata, expo, n_expo, n2 + 2 * fade, n2, fade, cum)
    assert np.array_equal(T, g["co_T_out"])  # the taper of T, float32 in place
    assert np.allclose(Tsum_stamp, g["co_Tsum_stamp"], rtol=1e-13, atol=0)
    assert np.allclose(Tsum_inpix, g["co_Tsum_inpix"], rtol=1e-12, atol=1e-15)
    assert np.allclose(Neff, g["co_Neff"], rtol=1e-11, atol=0)
    assert outimage.dtype == g["co_outimage"].dtype
    assert np.abs(outimage - g["co_outimage"]).max() <= 4e-7 * np.abs(g["co_outimage"]).max()

    with np.errstate(all="ignore"):
        for coef, dt, tag in ((-5000, np.uint16, "u5000"), (-10000, np.int16, "i10000"), (200000, np.int16, "i200000"), (50000, np.uint16, "u50000")):
            got = orc.compress_map(g["cmp_in"], coef, dt)
            assert got.dtype == g[f"cmp_{tag}"].dtype and np.array_equal(got, g[f"cmp_{tag}"]), tag


def test_block_accumulation_golden(golden):
    """Block._output_stamp_wrapper (coadd.py:1975-1993) and the boundary recovery of build_output_file (2163-2181)
    against the reference's own code run on nine finished stamps (make_golden_block.py): bit for bit."""
    g = golden("block_maps")
    n1P, n2, fk, n_out, n_inframe, n_inimage = (int(v) for v in g["pars"])
    ns = n1P * n2 + 2 * fk
    maps = {k: np.zeros((n_out, ns, ns), np.float32) for k in ("UC", "Sigma", "kappa", "Tsum", "Neff")}
    out_map = np.zeros((n_out, n_inframe, ns, ns), np.float32)
    src = dict(UC="UC", Sigma="Sigma", kappa="kappa", Tsum="Tsum_inpix", Neff="Neff")
    for j in range(1, n1P + 1):
        for i in range(1, n1P + 1):
            orc.block_accumulate(out_map, g[f"st{j}{i}_outimage"], j, i, n2, fk)
            for k, key in src.items():
                orc.block_accumulate(maps[k], g[f"st{j}{i}_{key}"], j, i, n2, fk)
    assert np.array_equal(out_map, g["acc_out_map"])
    for k in maps:
        assert np.array_equal(maps[k], g[f"acc_{k}_map"]), k
    orc.trapezoid_recover(out_map, fk)
    w = int(g["postage_pad"]) * n2
    pads = tuple(w * (s not in str(g["pad_sides"])) for s in "BTLR")
    assert np.array_equal(out_map, g["fin_out_map"])
    for k in maps:
        orc.trapezoid_recover(maps[k], fk, pads)
        assert np.array_equal(maps[k], g[f"fin_{k}_map"]), k


def _chain_inputs(g):
    """Geometry, per-group sampled PSFs and the nine InStamps of the output stamp in tests/golden/stamp_chain.npz."""
    n1P, n2, fade, n_inimage, n_inframe = (int(v) for v in g["pars"])
    geo = orc.Geom(int(g["npixpsf"]), int(g["oversamp"]), float(g["dtheta_as"]) / 3600.0, float(g["flat_penalty"]))
    ns, nst = geo.nsamp, n1P + 2
    lin = np.arange(ns) - (ns - 1) / 2.0
    gx, gy = np.meshgrid(lin, lin)
    xy = np.stack([gx.ravel(), gy.ravel()], axis=1) * geo.dscale  # psfutil.py:751-771
    inst = {(j, i): (g[f"in{j}{i}_x"], g[f"in{j}{i}_y"], g[f"in{j}{i}_data"], g[f"in{j}{i}_cum"].astype(np.int64)) for j in range(nst) for i in range(nst)}
    group_psfs, group_expo = {}, {}
    for gj in range(nst // 2):
        for gi in range(nst // 2):
            used = np.zeros(n_inimage, bool)
            for dj in range(2):
                for di in range(2):
                    used |= np.diff(inst[(2 * gj + dj, 2 * gi + di)][3]) > 0  # psfutil.py:812-818
            p0 = np.array([2 * gi * n2 - 0.5, 2 * gj * n2 - 0.5])          # coadd.py:712
            arr = []
            for e in np.flatnonzero(used):
                M, t0 = g[f"inM{e}"], g[f"int0{e}"]
                d = ((xy + p0) @ M.T + t0 - (p0[None, :] @ M.T + t0)) * geo.oversamp
                yxco = np.stack([d[:, 1].reshape(ns, ns), d[:, 0].reshape(ns, ns)])
                arr.append(orc.sample_psf(g[f"inpsf{e}"], ns, yxco))
            group_psfs[(gj, gi)] = orc.finish_psf_group(np.stack(arr), True, True)
            group_expo[(gj, gi)] = [int(e) for e in np.flatnonzero(used)]
    return geo, inst, group_psfs, group_expo, (n1P, n2, fade, n_inimage, n_inframe)


@pytest.mark.parametrize("name", ["stamp_chain", "stamp_chain_mid"])
def test_full_chain_golden(golden, name):
    """ONE output stamp end to end against the reference's own chain (make_golden_chain.py: PSFGrp sampling -> PSFOvl ->
    SysMatA / SysMatB -> OutStamp._build_system_matrices -> CholKernel -> taper -> _perform_coaddition, four PSF groups,
    one lacking an exposure; N = 29 and N = 220 input pixels).  The oracle starts from the same raw inputs (PSF images,
    affine maps, InStamp pixels)."""
    g = golden(name)
    geo, inst, group_psfs, group_expo, (n1P, n2, fade, n_inimage, n_inframe) = _chain_inputs(g)
    assert group_expo[(0, 1)] == [0, 2]
    ns, nst, n2f = geo.nsamp, n1P + 2, n2 + 2 * fade
    j_st, i_st = int(g["j_st"]), int(g["i_st"])
    tgt = orc.sample_psf(orc.get_outpsf("GAUSSIAN", 1.1, 2, ns, geo.oversamp), ns, None)[None]
    rft_out = orc.pad_and_rfft2(orc.finish_psf_group(tgt, True, True), geo)
    C = orc.overlap_out_C(rft_out, geo)
    assert np.allclose(C, g["C"], rtol=1e-14, atol=0)
    rft_in = {k: orc.pad_and_rfft2(v, geo) for k, v in group_psfs.items()}
    rpix = float(g["instamp_pad_as"]) / float(g["dtheta_as"])
    bottom, left = (j_st - 1) * n2, (i_st - 1) * n2
    top, right = bottom + n2 - 1, left + n2 - 1
    nine, piv, groups = [], [], []
    for dj in (-1, 0, 1):
        for di in (-1, 0, 1):
            nine.append(inst[(j_st + dj, i_st + di)])
            piv.append(([left - 0.5, None, right + 0.5][di + 1], [bottom - 0.5, None, top + 0.5][dj + 1]))
            groups.append(((j_st + dj) >> 1, (i_st + di) >> 1))
    sels = [orc.select_pixels(t[0], t[1], pv, rpix) for t, pv in zip(nine, piv)]
    x, y, indata, expo, cum = orc.process_input_stamps(nine, piv, rpix)
    assert np.array_equal(cum, g["inpix_cumsum"])
    ox, oy = left - fade + np.arange(n2f, dtype=np.float64), bottom - fade + np.arange(n2f, dtype=np.float64)
    A, mB = orc.stamp_system_groups(nine, sels, groups, rft_in, rft_out, geo, ox, oy, group_expo)
    assert np.abs(A - g["A"]).max() <= 1e-14 * np.abs(g["A"]).max()
    assert np.abs(mB - g["mBhalf"][0]).max() <= 1e-14 * np.abs(g["mBhalf"]).max()
    T, UC, Sg, kp, _ = orc.chol_kernel(g["A"], g["mBhalf"][0], float(g["C"][0]), g["kappaC"], 1e-6, 0.5)
    assert np.abs(T - g["T_raw"][0]).max() <= 1e-6 * np.abs(g["T_raw"]).max()
    s2 = (n2f, n2f)
    UC, Sg, kp = UC.reshape(s2).copy(), Sg.reshape(s2).copy(), kp.reshape(s2).copy()
    for a in (kp, Sg, UC):
        orc.trapezoid(a, fade)
    assert np.allclose(UC, g["UC"][0], rtol=2e-5, atol=1e-12) and np.allclose(Sg, g["Sigma"][0], rtol=2e-5) and np.array_equal(kp, g["kappa"][0])
    T3 = g["T_raw"].copy()
    outimage, Tsum_stamp, Tsum_inpix, Neff = orc.perform_coaddition(T3, indata, expo, n_inimage, n2f, n2, fade, cum)
    assert np.array_equal(T3, g["T"])
    # float32 einsum over N terms: numpy picks its reduction (BLAS or not) by operand layout, a few float32 ulps of the sum
    assert np.abs(outimage - g["outimage"]).max() <= 2e-6 * np.abs(g["outimage"]).max()
    assert np.allclose(Tsum_stamp, g["Tsum_stamp"], rtol=1e-12) and np.allclose(Neff, g["Neff"], rtol=1e-10)


@pytest.mark.parametrize("name", ["stamp_chain", "stamp_chain_mid"])
def test_chain_other_kernels_golden(golden, name):
    """The other LA kernels (Eigen one / two nodes, Cholesky three nodes, Iterative, Empirical) run by the reference on
    the chain's own A, -B/2, C (a real PSF-overlap system, not a synthetic one): oracle against those outputs."""
    g = golden(name)
    A, mB, C = g["A"], g["mBhalf"][0], float(g["C"][0])
    oy, ox = g["yx_val"][0].ravel().astype(np.float64), g["yx_val"][1].ravel().astype(np.float64)
    rho = float(g["instamp_pad_as"]) / float(g["dtheta_as"])
    runs = {"eig1": lambda kC: orc.eigen_kernel(A, mB, C, kC, 1e-6, 0.5), "eig2": lambda kC: orc.eigen_kernel(A, mB, C, kC, 1e-6, 0.5),
            "chol3": lambda kC: orc.chol_kernel(A, mB, C, kC, 1e-6, 0.5),
            "iter1": lambda kC: orc.iter_kernel(A, mB, C, kC, 1e-6, 0.5, oy, ox, g["iny_val"], g["inx_val"], rho),
            "emp": lambda kC: orc.empir_kernel(A, mB, C, kC, oy, ox, g["iny_val"], g["inx_val"], rho)}
    lam = np.linalg.eigvalsh(A)
    for tag, f in runs.items():
        kC = g[f"{tag}_kappaC"]
        T, UC, Sg, kp, _ = f(kC)
        kap = float(kC[0]) * C
        cond = (lam[-1] + kap) / (max(lam[0], 0.0) + kap)
        tT = 3e-5 if tag == "iter1" else 1e-6 + 100 * cond * 2.2e-16
        assert np.abs(T - g[f"{tag}_T"][0]).max() <= tT * np.abs(g[f"{tag}_T"]).max(), tag
        rt = 2e-3 if tag == "iter1" else 2e-5 + 100 * cond * 2.2e-16
        assert np.allclose(kp, g[f"{tag}_kappa"][0].ravel(), rtol=rt, atol=0), tag
        assert np.allclose(Sg, g[f"{tag}_Sigma"][0].ravel(), rtol=rt, atol=1e-9), tag
        assert np.allclose(UC, g[f"{tag}_UC"][0].ravel(), rtol=rt, atol=2e-7), tag


def test_iterative_clamp_golden(golden):
    """coadd.py:1104-1107 executed by the reference (make_golden_clamp.py): negatives, zeros, sub-1e-32 values and
    subnormals rise to float32(1e-32), NaN and +inf stay, dtype float32; other kernels leave the maps alone."""
    g = golden("iter_clamp")
    UC, Sigma = orc.iterative_clamp(g["UC_in"], g["Sigma_in"])
    assert UC.dtype == g["UC_Iterative"].dtype == np.float32
    assert np.array_equal(UC, g["UC_Iterative"], equal_nan=True) and np.array_equal(Sigma, g["Sigma_Iterative"], equal_nan=True)
    assert np.array_equal(g["UC_Cholesky"], g["UC_in"], equal_nan=True)
    assert np.nanmin(UC) == np.float32(1e-32) and np.isnan(UC).sum() == np.isnan(g["UC_in"]).sum()


@pytest.mark.parametrize("case", ["whole", "inner", "stop"])
def test_block_loop_order_golden(golden, case):
    """The reference's stamp loop statement (coadd.py:2056-2081, executed by tests/golden/make_golden_block_loop.py) against the
    oracle's restatement of its order + block_accumulate: visited stamps and every map bit for bit (float32 sums of three or
    four overlapping stamps of mixed magnitude: the order shows)."""
    g = golden("block_loop")
    n1P, n2, fk, n_out, n_inframe, n_inimage = (int(v) for v in g["pars"])
    lo_j, hi_j, lo_i, hi_i = (int(v) for v in g[f"{case}_window"])
    ids = orc.stamp_loop_order(lo_j, hi_j, lo_i, hi_i, int(g[f"{case}_nrun"]))
    assert np.array_equal(np.array(ids), g[f"{case}_visited"])
    ns = n1P * n2 + 2 * fk
    out_map = np.zeros((n_out, n_inframe, ns, ns), np.float32)
    names = dict(UC="UC_map", Sigma="Sigma_map", kappa="kappa_map", Tsum_inpix="Tsum_map", Neff="Neff_map")
    maps = {k: np.zeros((n_out, ns, ns), np.float32) for k in names}
    for j, i in ids:
        orc.block_accumulate(out_map, g[f"st{j}{i}_outimage"], j, i, n2, fk)
        for k in names:
            orc.block_accumulate(maps[k], g[f"st{j}{i}_{k}"], j, i, n2, fk)
    assert np.array_equal(out_map, g[f"{case}_out_map"])
    for k, nm in names.items():
        if f"{case}_{nm}" in g.files:
            assert np.array_equal(maps[k], g[f"{case}_{nm}"]), k
    rows = np.zeros_like(out_map)
    for j, i in sorted(ids):
        orc.block_accumulate(rows, g[f"st{j}{i}_outimage"], j, i, n2, fk)
    assert case != "whole" or not np.array_equal(rows, out_map)  # row by row is another sum


def test_iter_default_oracle_against_itself():
    """The parity statement for the Iterative kernel at the reference's default configuration (kappa = 0: tests/parity.py iter_parity,
    used by tests/test_gpu_iter_default.py for device against oracle) holds for the oracle against ITSELF with every selection in reverse
    pixel order -- the same conjugate-gradient recurrences (lakernel.py:397-442) with their sums in another order -- and is not
    tighter than that: ~2 % of the output pixels stop a step apart, where the steps agree T differs by up to 5e-4 of its largest entry."""
    from pyimcom_amd import synth
    from tests.parity import iter_parity

    cfg = synth.CONFIGS["iter_default"]
    psfs, target = synth.make_psfs(cfg, cfg.n_expo)
    g, tabs_ref, C_ref = orc.stamp_tables(cfg, psfs, target)
    E, C = cfg.n_expo, float(C_ref[0])
    tri = lambda i, j: (2 * E - i + 1) * i // 2 + j - i  # noqa: E731
    tab = np.array([[tri(a, b) if a <= b else (tri(b, a) | (1 << 30)) for b in range(E)] for a in range(E)], dtype=np.int32)
    st = synth.make_stamp(cfg, 0)
    A, Bt = orc.stamp_system(g, st.x, st.y, st.expo, tabs_ref, tab, np.zeros((E, E)), np.arange(E) + E * (E + 1) // 2, st.out_x0, st.out_y0, cfg.n2f)
    mB = np.ascontiguousarray(Bt.T)
    g1 = np.arange(cfg.n2f, dtype=np.float64)
    oy, ox = np.repeat(st.out_y0 + g1, cfg.n2f), np.tile(st.out_x0 + g1, cfg.n2f)
    args = (C, np.array(cfg.kappaC), cfg.uctarget, cfg.sigmamax, oy, ox)
    s1, s2 = [], []
    T1, U1, S1, k1, _ = orc.iter_kernel(A, mB, *args, st.y, st.x, cfg.rho, cfg.iter_rtol, cfg.iter_max, steps=s1)
    perm = np.arange(st.n)[::-1]
    Tp, U2, S2, k2, _ = orc.iter_kernel(A[np.ix_(perm, perm)], mB[:, perm], *args, st.y[perm], st.x[perm], cfg.rho, cfg.iter_rtol, cfg.iter_max, steps=s2)
    T2 = np.empty_like(Tp)
    T2[:, perm] = Tp
    relevant = orc._relevant(oy, ox, st.y, st.x, cfg.rho)
    rep = iter_parity(A, mB, C, relevant, cfg.iter_rtol, cfg.iter_max, (T2, np.array(s2), U2, S2), (T1, np.array(s1), U1, S1))
    assert 0.9 < rep["same_steps"] < 1.0 and rep["dT_same_max"] > 1e-5, rep  # (the statement is not vacuous: the two runs do differ)
    assert np.array_equal(k1, k2) and not k1.any()  # kappa = 0
    assert np.linalg.eigvalsh(A)[0] < 1e-10 * np.abs(A).max()  # the system the recurrences run on is singular to rounding
