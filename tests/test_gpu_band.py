"""The band basis of the Eigen path (csrc/band.hip): Householder reduction A = Q B Q^T to bandwidth 4 with lazy rank-2k
updates, against numpy -- the reflectors are multiplied out on the host, so Q^T A Q is checked entry by entry."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dense_band(band, n):
    B = np.zeros((n, n))
    for t in range(band.shape[0]):
        for i in range(n - t):
            B[i + t, i] = B[i, i + t] = band[t, i]
    return B


def _Q(V, tau, n):
    Q = np.eye(n)
    for r in range(n):
        if tau[r] != 0.0:
            v = V[r, :n]
            Q -= tau[r] * np.outer(Q @ v, v)  # Q <- Q H_r
    return Q


def test_band_reduce_vs_numpy_ragged_batch():
    """Matrices of 300, 193, 129, 70, 6, 5, 1 and 0 rows in one batch of ld = 384 (several lazy super-panels of 64 reflectors, a
    trailing GEMM update, groups cut by the matrix end, matrices too small for any reflector): band entries, zero fill outside
    the band, orthogonality of Q and the eigenvalues."""
    from pyimcom_amd.linalg import band_reduce

    rng = np.random.default_rng(4)
    ld = 384
    ns = [300, 193, 129, 70, 6, 5, 1, 0]
    A = np.zeros((len(ns), ld, ld))
    for s, n in enumerate(ns):
        X = rng.standard_normal((n, n))
        A[s, :n, :n] = X @ X.T / max(n, 1) + 0.01 * np.eye(n)
    A[:, np.arange(ld), np.arange(ld)] += (np.arange(ld)[None, :] >= np.array(ns)[:, None]) * 1.0  # identity padding, as the builders leave it
    band, V, tau = band_reduce(A, ns)
    for s, n in enumerate(ns):
        if n == 0:
            continue
        As = A[s, :n, :n]
        Q = _Q(V[s], tau[s], n)
        B = _dense_band(band[s][:, :n], n)
        M = Q.T @ As @ Q
        scale = np.abs(As).max()
        assert np.abs(Q.T @ Q - np.eye(n)).max() < 1e-13, s
        assert np.abs(np.triu(M, 5)).max() < 1e-13 * scale, s          # nothing outside the band
        assert np.abs(M - B).max() < 1e-13 * scale, (s, np.abs(M - B).max())
        assert np.abs(np.linalg.eigvalsh(B) - np.linalg.eigvalsh(As)).max() < 1e-13 * scale
        assert np.all(V[s][:, :4] == 0) and all(np.all(V[s][r, : r + 4] == 0) for r in range(n))  # zero above the pivots
        assert np.all(band[s][:, n:] == 0)


def test_symv4_forms_agree(monkeypatch):
    """The strip pass on the vector unit (the default) and on the matrix pipe (IMCOM_SYMV4=mfma, profiles/r05_negative_results.txt
    item 1): same band to rounding, each against LAPACK's eigenvalues, on a ragged batch that has full 64-row strips, strips cut by
    the matrix end and a matrix below one strip."""
    from pyimcom_amd.linalg import band_reduce

    rng = np.random.default_rng(41)
    ld = 640
    ns = [640, 577, 333, 64, 37]
    A = np.zeros((len(ns), ld, ld))
    for s, n in enumerate(ns):
        X = rng.standard_normal((n, n))
        A[s, :n, :n] = X @ X.T / n + 0.01 * np.eye(n)
    A[:, np.arange(ld), np.arange(ld)] += (np.arange(ld)[None, :] >= np.array(ns)[:, None]) * 1.0
    out = {}
    for form in ("valu", "mfma"):
        monkeypatch.setenv("IMCOM_SYMV4", form)
        out[form] = band_reduce(A.copy(), ns)[0]
    for s, n in enumerate(ns):
        scale = np.abs(A[s]).max()
        ev = np.linalg.eigvalsh(A[s, :n, :n])
        for form in out:
            assert np.abs(np.linalg.eigvalsh(_dense_band(out[form][s][:, :n], n)) - ev).max() < 1e-13 * scale, (form, s)
        # (different summation orders: the reflectors differ in rounding, and so do the bands -- signs included only through v's rounding)
        assert np.abs(np.abs(out["valu"][s]) - np.abs(out["mfma"][s])).max() < 1e-11 * scale, s


def test_band_reduce_on_cfg3_matrix():
    """The A of a cfg-3 stamp (N ~ 2.9k, ld = 2944) built on the device: the band's eigenvalues against LAPACK's of A, 2e-14 |A|."""
    import ctypes as C

    import torch
    from scipy.linalg import eigvals_banded

    from pyimcom_amd import synth
    from pyimcom_amd._lib import MEM_DEVICE, check, lib
    from pyimcom_amd.stamps import PSFGroupTables, StampBatch

    cfg = synth.CONFIGS["cfg3"]
    st = synth.make_stamp(cfg, 3)
    psfs, target = synth.make_psfs(cfg, st.n_expo)
    sb = StampBatch(cfg, [st], PSFGroupTables(psfs, target, cfg.nfft))
    sb.build()
    torch.cuda.synchronize()
    ld, n = sb.ldn, st.n
    band = torch.zeros((1, 5, ld), dtype=torch.float64, device="cuda:0")
    V = torch.zeros((1, ld, ld), dtype=torch.float64, device="cuda:0")
    tau = torch.zeros((1, ld), dtype=torch.float64, device="cuda:0")
    ns = np.array([n], dtype=np.int32)
    sb._stream()
    p = lambda t: C.c_void_p(t.data_ptr())
    check(lib.imcom_band_reduce(sb.ctx.handle, 1, ns.ctypes.data_as(C.c_void_p), ld, p(sb.A), p(band), p(V), p(tau), MEM_DEVICE))
    torch.cuda.synchronize()
    Ah = sb.A[0, :n, :n].cpu().numpy()
    w = np.linalg.eigvalsh(Ah)
    wb = eigvals_banded(band[0, :, :n].cpu().numpy(), lower=True)
    norm = max(abs(w[0]), abs(w[-1]))
    print(f"[band n={n}] |A|={norm:.3g} max |dlam|/|A| = {np.abs(wb - w).max() / norm:.2e}")
    assert np.abs(wb - w).max() <= 2e-14 * norm


def test_band_reduce_wide_matrix_generic_step_kernel():
    """ld = 3200 > 3072: the step kernel with the panel in LDS (band_step_kernel) instead of the register-resident one; the band's
    eigenvalues against LAPACK's of A and a sample of Q's columns for orthogonality."""
    from scipy.linalg import eigvals_banded

    from pyimcom_amd.linalg import band_reduce

    rng = np.random.default_rng(11)
    ld, n = 3200, 3125
    X = rng.standard_normal((n, 400))
    A = np.zeros((1, ld, ld))
    A[0, :n, :n] = X @ X.T / 400 + 0.05 * np.eye(n)
    A[0, np.arange(n, ld), np.arange(n, ld)] = 1.0
    band, V, tau = band_reduce(A, [n])
    w = np.linalg.eigvalsh(A[0, :n, :n])
    wb = eigvals_banded(band[0][:, :n], lower=True)
    assert np.abs(wb - w).max() <= 2e-14 * w[-1]
    # Q e_j for a few j: apply the reflectors in reverse order to unit vectors
    E = np.zeros((n, 3))
    E[[0, n // 2, n - 1], [0, 1, 2]] = 1.0
    for r in range(n - 1, -1, -1):
        if tau[0][r] != 0.0:
            v = V[0][r, :n]
            E -= tau[0][r] * np.outer(v, v @ E)
    assert np.abs(E.T @ E - np.eye(3)).max() < 1e-13


def test_eigen_kernel_with_and_without_overlap(monkeypatch):
    """c = Q^T b formed on the second stream while the reduction runs (default up to 128 stamps) or after it."""
    from tests.golden.make_golden import make_outst

    from pyimcom_amd.lakernel import HipEigenKernel

    rng = np.random.default_rng(5)
    n, m = 700, 256
    pts = rng.uniform(0, 30, (n, 2)); outp = rng.uniform(3, 27, (m, 2))
    A = np.exp(-((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1) / 3.0)
    B = np.exp(-((outp[:, None, :] - pts[None, :, :]) ** 2).sum(-1) / 3.5)[None]
    res = []
    for ov in ("0", "1"):
        monkeypatch.setenv("IMCOM_EIGEN_OVERLAP", ov)
        o = make_outst(A.copy(), B.copy(), np.array([1.0]), 16, np.array([1e-5, 1e-4, 1e-3]), 1e-6, 0.5)
        HipEigenKernel(o)()
        res.append(o)
    # both against the oracle's eigendecomposition kernel (N = 700: six panels of reflectors, i.e. three pairs; single and several kappa nodes
    # are covered by the golden cases at small N and by cfg-3 at N = 2.9k, 23 panels)
    from oracle import oracle as orc

    To, Uo, So, ko = orc.eigen_kernel(A.copy(), np.ascontiguousarray(B[0]), 1.0, np.array([1e-5, 1e-4, 1e-3]), 1e-6, 0.5)[:4]
    for o in res:
        flips = np.abs(o.kappa.ravel() / ko - 1.0) > 1e-5  # a bisection decision taken the other way (ties of the U/C target only)
        assert flips.sum() <= 2, int(flips.sum())
        ok = ~flips
        assert np.abs(o.T[0][ok] - To[ok]).max() <= 2e-6 * np.abs(To).max()
        assert np.allclose(o.UC.ravel()[ok], Uo[ok], rtol=1e-5, atol=1e-9) and np.allclose(o.Sigma.ravel()[ok], So[ok], rtol=1e-5, atol=1e-9)
    # (the overlapped pass applies the panels one by one as they are finished, the other one in pairs: rounding only)
    assert np.abs(res[0].T - res[1].T).max() <= 1e-6 * np.abs(res[1].T).max()
    for name in ("UC", "Sigma", "kappa"):
        assert np.allclose(getattr(res[0], name), getattr(res[1], name), rtol=1e-5, atol=1e-9), name
