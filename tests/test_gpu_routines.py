"""GPU parity, native-routine seam and LA-kernel seam: libimcom_hip (through the C-ABI) vs the oracle
and vs the golden vectors generated from the reference.  Run with -m gpu on an MI355X."""

import numpy as np
import pytest

from tests.golden.make_golden import cosine_system, gaussian_system, make_outst

pytestmark = pytest.mark.gpu

# Floating-point tolerances (fp64 internals, same interpolation as the reference):
TOL_INTERP = 1e-12  # |GPU - oracle| / max|table|: FMA contraction and nothing else
TOL_T = 1e-6        # max|dT| <= TOL_T * max|T| after the float32 cast
RTOL_MAP, ATOL_MAP = 1e-5, 1e-9  # UC, Sigma


@pytest.fixture(scope="module")
def hip():
    from pyimcom_amd import routines

    return routines


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle

    return oracle


def test_getw(hip, golden):
    g = golden("getw")
    for fh, w_ref in zip(g["fh"], g["w"]):
        w = np.zeros(10)
        hip.iD5512C_getw(w, float(fh))
        assert np.abs(w - w_ref).max() < 5e-16
    w = np.zeros(10)
    hip.iD5512C_getw(w, 0.5)  # reference tests/pyimcom/test_psf.py:55-61
    assert abs(w[5] - 1) < 1e-8 and np.abs(np.delete(w, 5)).max() < 1e-8


def test_interp_golden(hip, golden):
    g = golden("interp")
    f = np.full_like(g["f_scatter"], -7.0)
    hip.iD5512C(g["infunc"], g["x"], g["y"], f)
    assert np.array_equal(f == -7.0, g["f_scatter"] == -7.0)  # off-grid points untouched
    assert np.abs(f - g["f_scatter"]).max() < TOL_INTERP
    f = np.full_like(g["f_sym"], -7.0)
    hip.iD5512C_sym(g["infunc"], g["xs"], g["ys"], f)
    assert np.abs(f - g["f_sym"]).max() < TOL_INTERP
    f = np.full_like(g["f_grid"], -7.0)
    hip.gridD5512C(np.ascontiguousarray(g["infunc"][0]), g["xpos"], g["ypos"], f)
    assert np.abs(f - g["f_grid"]).max() < TOL_INTERP
    f = np.zeros_like(g["f_scatter_b"])
    hip.iD5512C(g["infunc_b"], g["xb"], g["yb"], f)
    assert np.abs(f - g["f_scatter_b"]).max() < TOL_INTERP * np.abs(g["infunc_b"]).max() * 10


def test_interp_large_vs_oracle(hip, orc):
    rng = np.random.default_rng(11)
    tab = rng.standard_normal((2, 395, 395))
    n = 200_000
    x = rng.uniform(-3.0, 400.0, n)  # includes off-grid points on every side
    y = rng.uniform(-3.0, 400.0, n)
    a, b = np.full((2, n), 3.25), np.full((2, n), 3.25)
    hip.iD5512C(tab, x, y, a)
    orc.iD5512C(tab, x, y, b)
    assert np.array_equal(a == 3.25, b == 3.25) and np.abs(a - b).max() < 5e-12
    # symmetric form: 300 x 300 positions, symmetric by construction
    sq = 300
    xs = rng.uniform(4.0, 389.0, (sq, sq)); xs = np.triu(xs) + np.triu(xs, 1).T
    ys = rng.uniform(4.0, 389.0, (sq, sq)); ys = np.triu(ys) + np.triu(ys, 1).T
    a, b = np.zeros((2, sq * sq)), np.zeros((2, sq * sq))
    hip.iD5512C_sym(tab, xs.ravel().copy(), ys.ravel().copy(), a)
    orc.iD5512C_sym(tab, xs.ravel().copy(), ys.ravel().copy(), b)
    assert np.abs(a - b).max() < 5e-12
    assert np.array_equal(a.reshape(2, sq, sq), a.reshape(2, sq, sq).transpose(0, 2, 1))
    # grid form at the cfg-2 geometry: 48 x 48 outputs at 1.894 table px pitch, plus an off-grid pixel
    npi, n2f = 37, 48
    x0 = rng.uniform(120.0, 280.0, npi); y0 = rng.uniform(120.0, 280.0, npi)
    x0[3] = 380.0; y0[5] = -20.0
    y0[7] = 40.0; x0[9] = 50.0; y0[11] = 420.0  # grids that leave the table part-way (rows / columns of zero weight)
    pitch = 1.894
    xp = np.ascontiguousarray(x0[:, None] - pitch * np.arange(n2f)[None, :])
    yp = np.ascontiguousarray(y0[:, None] - pitch * np.arange(n2f)[None, :])
    a, b = np.zeros((npi, n2f * n2f)), np.zeros((npi, n2f * n2f))
    hip.gridD5512C(tab[0], xp, yp, a)
    orc.gridD5512C(np.ascontiguousarray(tab[0]), xp, yp, b)
    assert np.abs(a - b).max() < 5e-12
    # irregular (non-monotonic, widely spread) positions force the direct-evaluation fallback
    xp = np.ascontiguousarray(rng.uniform(0.0, 395.0, (5, 40))); yp = np.ascontiguousarray(rng.uniform(0.0, 395.0, (5, 33)))
    a, b = np.zeros((5, 40 * 33)), np.zeros((5, 40 * 33))
    hip.gridD5512C(tab[1], xp, yp, a)
    orc.gridD5512C(np.ascontiguousarray(tab[1]), xp, yp, b)
    assert np.abs(a - b).max() < 5e-12


def test_lakernel1(hip, orc, golden):
    g = golden("lakernel1_small")
    lam, mP = np.ascontiguousarray(g["lam"]), np.ascontiguousarray(g["mPhalf"])
    m, n = mP.shape
    k, S, U, T = np.zeros(m), np.zeros(m), np.zeros(m), np.zeros((m, n))
    hip.lakernel1(lam, None, mP, 0.7, 1e-8, 1e-16, 1e16, 53, k, S, U, T, 0.5)
    # the reference's own C-vs-numba tolerances (tests/pyimcom/test_routine.py:133-145)
    assert np.abs(k - g["kappa"]).max() < 1e-12 and np.abs(S - g["Sigma"]).max() < 1e-7
    assert np.abs(U - g["UC"]).max() < 1e-13 and np.abs(T - g["T"]).max() < 1e-8
    # the full-size case of the reference test, with its known-answer windows
    A, mB, C = gaussian_system(33, 25)
    lam, Q = np.linalg.eigh(A)
    mP = np.ascontiguousarray(mB @ Q)
    m, n = mP.shape
    k, S, U, T = np.zeros(m), np.zeros(m), np.zeros(m), np.zeros((m, n))
    hip.lakernel1(lam, Q, mP, C, 1e-8, 1e-16, 1e16, 53, k, S, U, T, 0.5)
    k2, S2, U2, T2 = np.zeros(m), np.zeros(m), np.zeros(m), np.zeros((m, n))
    orc.lakernel1(lam, Q, mP, C, 1e-8, 1e-16, 1e16, 53, k2, S2, U2, T2, 0.5)
    assert np.abs(k - k2).max() < 1e-12 and np.abs(S - S2).max() < 1e-7 and np.abs(U - U2).max() < 1e-13
    assert np.abs(T - T2).max() < 1e-8
    assert 2.5e-7 < k.min() and k.max() < 3.5e-7 and 0.34 < S.min() and S.max() < 0.38
    assert 9e-9 < U.min() and U.max() < 1.1e-8 and 0.077 < np.abs(T).max() < 0.079


def test_build_reduced_T(hip, golden):
    g = golden("build_reduced_T")
    m, nv = g["brt_a_kappa"].size, g["kappa_nodes"].size
    for tag in "abc":
        ok, oS, oU, ow = np.zeros(m), np.zeros(m), np.zeros(m), np.zeros(m * nv)
        hip.build_reduced_T_wrap(g["Nflat"], g["Dflat"], g["Eflat"], g["kappa_nodes"], float(g[f"brt_{tag}_ucmin"]),
                                 float(g[f"brt_{tag}_smax"]), ok, oS, oU, ow)
        assert np.allclose(ok, g[f"brt_{tag}_kappa"], rtol=1e-12, atol=0)
        assert np.allclose(oS, g[f"brt_{tag}_Sigma"], rtol=1e-9, atol=1e-14)
        assert np.allclose(oU, g[f"brt_{tag}_UC"], rtol=0, atol=1e-12)
        assert np.allclose(ow, g[f"brt_{tag}_w"], rtol=1e-8, atol=1e-12)


# ------------------------------------------------------------------------------------------------ LA seam
def _run(K, A, mB, C, n2f, kC, uct, smax):
    outst = make_outst(A.copy(), mB.copy(), C, n2f, kC, uct, smax)
    k = K(outst)
    k()
    assert np.array_equal(outst.sysmata, A)  # caller-owned, unmodified
    return outst


CHOL_CASES = [("cos_chol1", "cos", [1e-2], 1e-4, 0.5, 4), ("cos_cholm", "cos", [1e-4, 1e-3, 1e-2], 1e-4, 1.0, 4),
              ("gau_chol1", "gau", [6e-4], 1e-6, 0.5, 9), ("gau_cholm", "gau", [1e-5, 1e-4, 1e-3], 1e-6, 0.5, 9)]


@pytest.mark.parametrize("name,sysn,kC,uct,smax,n2f", CHOL_CASES)
def test_chol_kernel_golden(golden, name, sysn, kC, uct, smax, n2f):
    from pyimcom_amd.lakernel import HipCholKernel

    g = golden("lakernel")
    o = _run(HipCholKernel, g[f"{sysn}_A"], g[f"{sysn}_mBhalf"], np.atleast_1d(g[f"{sysn}_C"]), n2f, np.array(kC), uct, smax)
    assert o.T.dtype == np.float32 and o.T.shape == g[f"{name}_T"].shape and o.UC.shape == g[f"{name}_UC"].shape
    assert np.abs(o.T - g[f"{name}_T"]).max() <= TOL_T * np.abs(g[f"{name}_T"]).max()
    assert np.allclose(o.UC, g[f"{name}_UC"], rtol=RTOL_MAP, atol=ATOL_MAP)
    assert np.allclose(o.Sigma, g[f"{name}_Sigma"], rtol=RTOL_MAP, atol=ATOL_MAP)
    assert np.allclose(o.kappa, g[f"{name}_kappa"], rtol=1e-6, atol=0)


def test_chol_kernel_known_answers():
    """Range assertions of the reference's tests/pyimcom/test_la.py:92-99 on the HIP Cholesky kernel
    (single kappa gives the same maps as the eigen kernel of that test)."""
    from pyimcom_amd.lakernel import HipCholKernel

    A, mB, C = cosine_system()
    o = _run(HipCholKernel, A, mB, np.array([C]), 4, [1e-2], 1e-4, 0.5)
    assert np.all(o.UC >= 0)
    for j in range(16):
        assert (o.UC.ravel()[j] < 1e-4) if j % 5 == 0 else (0.05 < o.UC.ravel()[j] < 0.2)
        assert 0.6 < o.Sigma.ravel()[j] < 1.0 and 0.002 < o.kappa.ravel()[j] < 0.004


def test_chol_empty_stamp(golden):
    from pyimcom_amd.lakernel import HipCholKernel

    g = golden("lakernel")
    o = _run(HipCholKernel, np.zeros((0, 0)), np.zeros((1, 16, 0)), np.array([1.0]), 4, np.array([1e-3]), 1e-4, 0.5)
    assert o.T.shape == (1, 16, 0) and o.T.dtype == np.float32
    assert np.array_equal(o.UC, g["empty_UC"]) and np.array_equal(o.Sigma, g["empty_Sigma"])
    assert np.array_equal(o.kappa, g["empty_kappa"])


@pytest.mark.parametrize("nv", [1, 3])
def test_chol_ragged_batch_vs_oracle(orc, nv):
    """Direct C-ABI call: three stamps of different N (one spanning several 128-blocks, one empty) in one batch."""
    import ctypes as C

    from pyimcom_amd._lib import MEM_HOST, check, default_context, lib

    rng = np.random.default_rng(7 + nv)
    ns = np.array([300, 0, 131], dtype=np.int32)
    ldn, m = 300, 150
    A = np.zeros((3, ldn, ldn)); B = np.zeros((3, m, ldn)); Cs = np.array([0.9, 1.0, 1.3])
    for s, n in enumerate(ns):
        if n == 0:
            continue
        pts = rng.uniform(0, 12, (n, 2)); outp = rng.uniform(2, 10, (m, 2))
        A[s, :n, :n] = Cs[s] * np.exp(-((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1) / 3.0)
        B[s, :, :n] = Cs[s] * np.exp(-((outp[:, None, :] - pts[None, :, :]) ** 2).sum(-1) / 3.5)
    kC = np.array([6e-4]) if nv == 1 else np.array([1e-5, 1e-4, 1e-3])
    T = np.zeros((3, m, ldn), np.float32); UC = np.zeros((3, m), np.float32)
    Sg = np.zeros((3, m), np.float32); kp = np.zeros((3, m), np.float32); info = np.zeros(3, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    check(lib.imcom_solve_chol(default_context().handle, 3, p(ns), ldn, m, p(A), p(B), p(Cs), p(kC), nv, 1e-6, 0.5,
                               p(T), p(UC), p(Sg), p(kp), p(info), MEM_HOST))
    for s, n in enumerate(ns):
        if n == 0:
            assert np.all(UC[s] == 1) and np.all(Sg[s] == 0) and np.all(kp[s] == 1) and np.all(T[s] == 0)
            continue
        To, Uo, So, ko, _ = orc.chol_kernel(A[s, :n, :n].copy(), np.ascontiguousarray(B[s, :, :n]), Cs[s], kC, 1e-6, 0.5)
        assert np.abs(T[s, :, :n] - To).max() <= TOL_T * np.abs(To).max()
        assert np.all(T[s, :, n:] == 0)
        assert np.allclose(UC[s], Uo, rtol=RTOL_MAP, atol=ATOL_MAP) and np.allclose(Sg[s], So, rtol=RTOL_MAP, atol=ATOL_MAP)
        assert np.allclose(kp[s], ko, rtol=1e-5, atol=0)


@pytest.mark.parametrize("nv", [1, 3])
def test_chol_stamps_seam_vs_oracle(orc, nv):
    """solve_chol_stamps: four OutStamps of different N (one empty, one spanning several 128-blocks), each with its own host arrays,
    in ONE call -- every output against the oracle and against the kernel called stamp by stamp as the reference does
    (coadd.py:1091-1093)."""
    from pyimcom_amd.lakernel import HipCholKernel, solve_chol_stamps

    rng = np.random.default_rng(21 + nv)
    ns, m, n2f = [300, 0, 131, 257], 144, 12
    kC = np.array([6e-4]) if nv == 1 else np.array([1e-5, 1e-4, 1e-3])
    outs, singles, sys_ = [], [], []
    for n in ns:
        Cc = float(rng.uniform(0.8, 1.3))
        pts = rng.uniform(0, 12, (n, 2)); outp = rng.uniform(2, 10, (m, 2))
        A = Cc * np.exp(-((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1) / 3.0)
        B = (Cc * np.exp(-((outp[:, None, :] - pts[None, :, :]) ** 2).sum(-1) / 3.5))[None]
        sys_.append((A, B, Cc))
        outs.append(make_outst(A.copy(), B.copy(), np.array([Cc]), n2f, kC, 1e-6, 0.5))
        singles.append(make_outst(A.copy(), B.copy(), np.array([Cc]), n2f, kC, 1e-6, 0.5))
    ks = solve_chol_stamps(outs)
    assert len(ks) == 4 and all(k.info.shape == (1,) for k in ks)
    for o, o1, (A, B, Cc), n in zip(outs, singles, sys_, ns):
        HipCholKernel(o1)()
        assert np.array_equal(o.sysmata, A)  # caller-owned, unmodified
        assert o.T.shape == (1, m, n) and o.T.dtype == np.float32 and o.UC.shape == (1, n2f, n2f)
        if n == 0:
            assert np.all(o.UC == 1) and np.all(o.Sigma == 0) and np.all(o.kappa == 1)
            continue
        To, Uo, So, ko, _ = orc.chol_kernel(A.copy(), np.ascontiguousarray(B[0]), Cc, kC, 1e-6, 0.5)
        assert np.abs(o.T[0] - To).max() <= TOL_T * np.abs(To).max()
        assert np.allclose(o.UC.ravel(), Uo, rtol=RTOL_MAP, atol=ATOL_MAP) and np.allclose(o.Sigma.ravel(), So, rtol=RTOL_MAP, atol=ATOL_MAP)
        assert np.allclose(o.kappa.ravel(), ko, rtol=1e-5, atol=0)
        # the batch only changes the order of the factorisation's partial sums
        assert np.abs(o.T - o1.T).max() <= TOL_T * np.abs(To).max() and np.allclose(o.kappa, o1.kappa, rtol=1e-5, atol=0)


# ------------------------------------------------------------------------------------------------ eigen path
def test_eigh_vs_numpy():
    from pyimcom_amd.linalg import eigh

    rng = np.random.default_rng(3)
    for n in (5, 64, 150, 300):
        M = rng.standard_normal((n, n))
        A = M + M.T
        lam, Q = eigh(A)
        w = np.linalg.eigvalsh(A)
        assert np.abs(lam - w).max() < 1e-12 * np.abs(w).max()
        assert np.abs(Q.T @ Q - np.eye(n)).max() < 1e-12
        assert np.abs(A @ Q - Q * lam).max() < 1e-11 * np.abs(w).max()
    # the numerically semi-definite Gaussian overlap matrix of the LA tests (eigenvalues down to ~1e-17)
    A, _, _ = gaussian_system(13, 9, sigma=2.0, off=4.0, step=0.5)
    lam, Q = eigh(A)
    w = np.linalg.eigvalsh(A)
    assert np.abs(lam - w).max() < 2e-14 * w[-1]
    assert np.abs(A @ Q - Q * lam).max() < 1e-13 * w[-1] * 10
    # batch with equal sizes
    As = np.stack([A, 2 * A + np.eye(A.shape[0])])
    lams, Qs = eigh(As)
    assert np.abs(lams[1] - (2 * w + 1)).max() < 1e-12 * (2 * w[-1] + 1)


EIG_CASES = [("cos_eig1", "cos", [1e-2], 1e-4, 0.5, 4), ("cos_eigm", "cos", [1e-4, 1e-3, 1e-2], 1e-4, 1.0, 4),
             ("gau_eig1", "gau", [6e-4], 1e-6, 0.5, 9), ("gau_eigm", "gau", [1e-5, 1e-4, 1e-3], 1e-6, 0.5, 9)]


@pytest.mark.parametrize("name,sysn,kC,uct,smax,n2f", EIG_CASES)
def test_eigen_kernel_golden(golden, name, sysn, kC, uct, smax, n2f):
    from pyimcom_amd.lakernel import HipEigenKernel

    g = golden("lakernel")
    o = _run(HipEigenKernel, g[f"{sysn}_A"], g[f"{sysn}_mBhalf"], np.atleast_1d(g[f"{sysn}_C"]), n2f, np.array(kC), uct, smax)
    assert o.T.dtype == np.float32 and o.T.shape == g[f"{name}_T"].shape
    assert np.abs(o.T - g[f"{name}_T"]).max() <= TOL_T * np.abs(g[f"{name}_T"]).max()
    assert np.allclose(o.UC, g[f"{name}_UC"], rtol=RTOL_MAP, atol=ATOL_MAP)
    assert np.allclose(o.Sigma, g[f"{name}_Sigma"], rtol=RTOL_MAP, atol=ATOL_MAP)
    assert np.allclose(o.kappa, g[f"{name}_kappa"], rtol=1e-6, atol=0)


def test_eigen_kernel_known_answers():
    """tests/pyimcom/test_la.py:46-160 (test_eigen, test_eigen2) on the HIP eigen kernel, incl. the kappa*C^2 quirk."""
    from pyimcom_amd.lakernel import HipEigenKernel

    A, mB, C = cosine_system()
    o = _run(HipEigenKernel, A, mB, np.array([C]), 4, [1e-2], 1e-4, 0.5)
    assert np.all(o.UC >= 0)
    for j in range(16):
        assert (o.UC.ravel()[j] < 1e-4) if j % 5 == 0 else (0.05 < o.UC.ravel()[j] < 0.2)
        assert 0.6 < o.Sigma.ravel()[j] < 1.0 and 0.002 < o.kappa.ravel()[j] < 0.004
    o = _run(HipEigenKernel, A, mB, np.array([C]), 4, [1e-4, 1e-3, 1e-2], 1e-4, 1.0)
    assert np.all(o.UC >= 0)
    for j in range(16):
        if j % 5 == 0:
            assert o.UC.ravel()[j] < 1e-4 and 5e-4 < o.kappa.ravel()[j] < 1.5e-3
        else:
            assert 0.05 < o.UC.ravel()[j] < 0.2 and 5e-6 < o.kappa.ravel()[j] < 1.5e-5
        assert 0.6 < o.Sigma.ravel()[j] < 1.0


def test_eigh_indefinite_pairs():
    """+x / -x eigenvalue pairs (degenerate for a plain one-sided Jacobi, which sees A^2) must be resolved."""
    from pyimcom_amd.linalg import eigh

    rng = np.random.default_rng(8)
    n = 96
    Qr, _ = np.linalg.qr(rng.standard_normal((n, n)))
    w = np.concatenate([-np.linspace(0.1, 2.0, n // 2), np.linspace(0.1, 2.0, n // 2)])
    A = (Qr * w) @ Qr.T
    A = 0.5 * (A + A.T)
    lam, Q = eigh(A)
    assert np.abs(lam - np.sort(w)).max() < 1e-13
    assert np.abs(A @ Q - Q * lam).max() < 1e-12


def test_chol_repair_golden(golden):
    """Non-positive-definite A + kappa I: the eigh-shift repair of lakernel.py:262-279 (golden 'repair_*',
    system of tests/pyimcom/test_la.py:8-24), with lambda_min computed on the GPU."""
    from pyimcom_amd.lakernel import HipCholKernel

    g = golden("lakernel")
    A6, mB6, C6 = cosine_system()
    o = _run(HipCholKernel, g["repair_A"], mB6, np.array([C6]), 4, np.array([1e-4 / C6]), 1e-4, 0.5)
    assert np.abs(o.T - g["repair_chol1_T"]).max() <= 1e-5 * np.abs(g["repair_chol1_T"]).max()  # cond ~ 5e3 after the shift
    assert np.allclose(o.UC, g["repair_chol1_UC"], rtol=1e-4, atol=1e-7)
    assert np.allclose(o.Sigma, g["repair_chol1_Sigma"], rtol=1e-4, atol=1e-7)
    assert np.allclose(o.kappa, g["repair_chol1_kappa"], rtol=1e-6, atol=0)


def test_chol_repair_multi_kappa_vs_oracle(orc):
    """Multi-kappa: only the first node needs the repair; later nodes continue from the restored diagonal."""
    import ctypes as C

    from pyimcom_amd._lib import MEM_HOST, check, default_context, lib

    A, mB, Cc = gaussian_system(11, 7, sigma=1.2, off=3.0, step=0.7)
    n, m = A.shape[0], mB.shape[0]
    A = A - (np.linalg.eigvalsh(A)[0] + 2e-4 * Cc) * np.eye(n)  # lambda_min = -2e-4 C: node 1e-5 fails, 1e-3 does not
    kC = np.array([1e-5, 1e-3, 1e-2])
    To, Uo, So, ko, info_o = orc.chol_kernel(A.copy(), mB, Cc, kC, 1e-6, 0.5)
    assert info_o == 1
    T = np.zeros((1, m, n), np.float32); UC = np.zeros((1, m), np.float32)
    Sg = np.zeros((1, m), np.float32); kp = np.zeros((1, m), np.float32); info = np.zeros(1, np.int32)
    ns = np.array([n], np.int32); Cs = np.array([Cc])
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    check(lib.imcom_solve_chol(default_context().handle, 1, p(ns), n, m, p(A), p(np.ascontiguousarray(mB)), p(Cs), p(kC), 3, 1e-6,
                               0.5, p(T), p(UC), p(Sg), p(kp), p(info), MEM_HOST))
    assert info[0] == 1
    assert np.abs(T[0] - To).max() <= 1e-4 * np.abs(To).max()  # the repaired node is singular to ~1e-16: cond ~ 1e12 there
    assert np.allclose(UC[0], Uo, rtol=1e-3, atol=1e-6) and np.allclose(Sg[0], So, rtol=1e-3, atol=1e-6)


def test_chol_repair_of_many_stamps_in_one_batch(orc):
    """A batch in which most factorisations fail (different matrices, different shifts, one healthy stamp in between and a
    shorter one): the smallest eigenvalues of the failed stamps are computed in groups by one eigensolver batch
    (lambda_min_group) and every stamp is repaired as the reference repairs it alone (lakernel.py:262-279)."""
    import ctypes as C

    from pyimcom_amd._lib import MEM_HOST, check, default_context, lib

    A0, mB0, Cc = gaussian_system(11, 7, sigma=1.2, off=3.0, step=0.7)
    n, m = A0.shape[0], mB0.shape[0]
    lam0 = np.linalg.eigvalsh(A0)[0]
    batch = 37  # more than one group of 32
    ns = np.full(batch, n, np.int32)
    ns[5] = n - 9  # a ragged stamp
    A = np.zeros((batch, n, n)); mB = np.zeros((batch, m, n)); Cs = np.full(batch, Cc)
    kC = np.array([1e-5, 1e-3])
    want = []
    for s in range(batch):
        k = int(ns[s])
        scale = 1.0 + 0.03 * s
        shift = 0.0 if s == 3 else (np.linalg.eigvalsh(A0[:k, :k])[0] + (1.5e-4 + 1e-5 * s) * Cc)  # stamp 3 needs no repair
        A[s, :k, :k] = scale * (A0[:k, :k] - shift * np.eye(k))
        mB[s, :, :k] = scale * mB0[:, :k]
        Cs[s] = scale * Cc
        want.append(orc.chol_kernel(A[s, :k, :k].copy(), mB[s, :, :k].copy(), Cs[s], kC, 1e-6, 0.5))
    T = np.zeros((batch, m, n), np.float32); UC = np.zeros((batch, m), np.float32)
    Sg = np.zeros((batch, m), np.float32); kp = np.zeros((batch, m), np.float32); info = np.zeros(batch, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    check(lib.imcom_solve_chol(default_context().handle, batch, p(ns), n, m, p(A), p(mB), p(Cs), p(kC), 2, 1e-6, 0.5, p(T), p(UC), p(Sg),
                               p(kp), p(info), MEM_HOST))
    assert info[3] == 0 and want[3][4] == 0
    for s in range(batch):
        To, Uo, So, ko, info_o = want[s]
        k = int(ns[s])
        assert int(info[s]) == info_o, s
        assert np.abs(T[s, :, :k] - To).max() <= 1e-4 * np.abs(To).max(), s
        assert np.all(T[s, :, k:] == 0)
        assert np.allclose(UC[s], Uo, rtol=1e-3, atol=1e-6) and np.allclose(Sg[s], So, rtol=1e-3, atol=1e-6) and np.allclose(kp[s], ko, rtol=1e-5), s


def test_croutines_shim_is_importable_as_top_level_module(golden):
    """The injection seam of the reference (lakernel.py:41-47, psfutil.py:37-49; tests/pyimcom/test_missing.py): with
    furry_parakeet absent, a top-level module named `pyimcom_croutines` on sys.path is picked up unchanged."""
    import importlib
    import os
    import sys

    import pyimcom_amd

    shim_dir = os.path.join(os.path.dirname(pyimcom_amd.__file__), "shim")
    sys.path.insert(0, shim_dir)
    try:
        sys.modules.pop("pyimcom_croutines", None)
        mod = importlib.import_module("pyimcom_croutines")
        for name in ("iD5512C", "iD5512C_sym", "gridD5512C", "lakernel1", "build_reduced_T_wrap"):
            assert callable(getattr(mod, name))
        g = golden("interp")
        out = np.full_like(g["f_scatter"], -7.0)  # the golden keeps -7 where the point is off the grid (output untouched)
        mod.iD5512C(g["infunc"], g["x"], g["y"], out)
        assert np.abs(out - g["f_scatter"]).max() < TOL_INTERP
    finally:
        sys.path.remove(shim_dir)
        sys.modules.pop("pyimcom_croutines", None)


@pytest.mark.gpu
def test_rate_probes_run():
    """The two diagnostics behind the roofline discussion run and return sane rates: a pure fp64 MFMA loop
    (imcom_ctx_mfma_probe) and the tile engine's k loop as a plain batched product, 128 x 128 against 256 x 128 tiles
    (imcom_ctx_gemm_probe)."""
    from pyimcom_amd._lib import default_context, lib

    ctx = default_context()
    peak = ctx.mfma_probe(10.0)
    assert 40.0 < peak < 90.0, peak
    for variant in (0, 1) if lib.imcom_dev_build() else (0,):  # the 256 x 128 probe kernel ships with the developer build only
        rate = ctx.gemm_probe(variant, 1024, 1024, 1024, 8, 2)
        assert 10.0 < rate < peak, (variant, rate, peak)
