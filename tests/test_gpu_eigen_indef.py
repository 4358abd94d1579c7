"""The Eigen kernel on system matrices that are not positive semi-definite.

The reference's EigenKernel (lakernel.py:154-223) diagonalises A and divides by lam + kappa whatever its sign: it is the kernel
a user turns to when the Cholesky factorisation fails.  The device path works in a band basis with an unpivoted LDL^T, which is
safe only where A + kappa I is positive definite; every stamp is therefore checked (all pivots at the lowest kappa of the call)
and a stamp that fails is re-solved through the eigendecomposition itself, reported with info = 1 (csrc/eigen.hip)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _O:
    pass


def _outst(A, mB, C, n2f, kC, uct, smax):
    o, o.blk = _O(), _O()
    o.blk.cfg = c = _O()
    c.n_out, c.n2f, c.kappaC_arr, c.uctarget, c.sigmamax = mB.shape[0], n2f, np.asarray(kC, dtype=np.float64), uct, smax
    o.sysmata, o.mhalfb, o.outovlc, o.inpix_cumsum = A, mB, np.atleast_1d(C), np.array([A.shape[0]])
    return o


@pytest.mark.parametrize("name", ["repair", "gau", "chain"])
@pytest.mark.parametrize("tag", ["eig1", "eigm"])
def test_eigen_kernel_class_on_indefinite_goldens(golden, name, tag):
    """HipEigenKernel (imcom_solve_eigen, host arrays) against the outputs of the reference's EigenKernel on matrices with
    eigenvalues below -kappa (tests/golden/eigen_indef.npz): 6 x 6, Gaussian N = 169 with two target PSFs, and a real
    PSF-overlap matrix of N = 220 with 133 negative eigenvalues; one kappa node and several."""
    from pyimcom_amd.lakernel import HipEigenKernel
    from tests.test_oracle import indef_case

    g, A, mB, C, n2f = indef_case(golden, name)
    kC = g[f"{name}_{tag}_kappaC"]
    o = _outst(A.copy(), mB, C, n2f, kC, float(g[f"{name}_uctarget"]), float(g[f"{name}_sigmamax"]))
    k = HipEigenKernel(o)
    k()
    assert np.array_equal(o.sysmata, A)  # the caller's matrix is not modified (lakernel.py contract)
    assert np.all(k.info == 1), k.info  # every target PSF's A + kappa_min I is indefinite here
    ref = {q: g[f"{name}_{tag}_{q}"] for q in ("T", "UC", "Sigma", "kappa")}
    # forward error of the eigen-sums ~ eps |A| / min |lam + kappa|: 3e-8 relative at worst in these fixtures (printed by the generator)
    assert np.abs(o.T - ref["T"]).max() <= 2e-6 * np.abs(ref["T"]).max()
    assert np.allclose(o.UC, ref["UC"], rtol=2e-5, atol=1e-9) and np.allclose(o.Sigma, ref["Sigma"], rtol=2e-5, atol=1e-9)
    assert np.allclose(o.kappa, ref["kappa"], rtol=1e-6, atol=0)


def test_eigen_kernel_class_positive_definite_keeps_the_band_path(golden):
    """The same systems unshifted are positive definite at kappa_min: info = 0 (the band-basis route), results as before."""
    from pyimcom_amd.lakernel import HipEigenKernel

    g = golden("lakernel")
    o = _outst(g["gau_A"].copy(), g["gau_mBhalf"], g["gau_C"], 9, [1e-5, 1e-4, 1e-3], 1e-6, 0.5)
    k = HipEigenKernel(o)
    k()
    assert np.all(k.info == 0)
    assert np.abs(o.T - g["gau_eigm_T"]).max() <= 2e-6 * np.abs(g["gau_eigm_T"]).max()


@pytest.mark.parametrize("kind", ["multi", "single"])
def test_eigen_resident_mixed_batch_with_indefinite_stamps_cfg3(kind):
    """cfg-3 at BASELINE size (8 exposures, N ~ 2.9k, m = 2304) through the resident entry, a batch of three stamps of which the
    first and the last have their A shifted by -2 kappa_max I after the build: A + kappa I is indefinite for every kappa of the
    bracket (hundreds of eigenvalues below -kappa_max, some inside the bracket).  One call serves both kinds: info = [1, 0, 1];
    the shifted stamps against orc.eigen_kernel on the shifted matrix (numpy eigh + the C lakernel1), the untouched one too."""
    import dataclasses

    import torch

    from oracle import oracle as orc
    from pyimcom_amd import synth
    from pyimcom_amd.stamps import PSFGroupTables, StampBatch

    cfg = synth.CONFIGS["cfg3"]
    if kind == "single":
        cfg = dataclasses.replace(cfg, kappaC=(cfg.kappaC[1],))
    stamps = [synth.make_stamp(cfg, 40 + i) for i in range(3)]
    psfs, target = synth.make_psfs(cfg, max(s.n_expo for s in stamps))
    tabs = PSFGroupTables(psfs, target, cfg.nfft)
    sb = StampBatch(cfg, stamps, tabs)
    sb.build()
    C = float(tabs.Cs[0])
    shift = 2.0 * cfg.kappaC[-1] * C
    for b in (0, 2):
        n = stamps[b].n
        sb.A[b, :n, :n] -= shift * torch.eye(n, dtype=torch.float64, device=sb.A.device)
    torch.cuda.synchronize()
    A_h = [sb.A[b, : st.n, : st.n].cpu().numpy() for b, st in enumerate(stamps)]
    B_h = [sb.Bt[b, : st.n, : cfg.m].cpu().numpy().T.copy() for b, st in enumerate(stamps)]
    sb.solve()
    torch.cuda.synchronize()
    res = sb.result()
    assert res.info.tolist() == [1, 0, 1]
    flips = 0
    for b, st in enumerate(stamps):
        lam = np.linalg.eigvalsh(A_h[b])
        kmin, kmax = cfg.kappaC[0] * C, cfg.kappaC[-1] * C
        if b != 1:
            assert lam[0] < -kmax and (lam < -kmax).sum() > 100
        T, UC, Sigma, kappa, _ = orc.eigen_kernel(A_h[b], B_h[b], C, cfg.kappaC, cfg.uctarget, cfg.sigmamax)
        k_gpu, k_ref = res.kappa[b].cpu().numpy().ravel().astype(np.float64), kappa.astype(np.float64)
        same = np.abs(k_gpu / k_ref - 1.0) <= 1e-6  # (a flipped bisection decision moves kappa by >= 2.8e-4: see test_cfg3_full_vs_oracle)
        flips += int((~same).sum())
        # distance of the evaluated kappas from the poles at -lam_i sets the conditioning of the eigen-sums
        kap_true = np.unique(k_ref / C if kind == "multi" else k_ref)  # (multi: the reference reports kappa * C, lakernel.py:222)
        gap = np.abs(lam[None, :] + kap_true[:, None]).min()
        tol = 1e-6 + 200 * 2.2e-16 * max(abs(lam[0]), lam[-1]) / max(gap, 1e-300)
        Tg = res.T(b).cpu().numpy()
        eT = np.abs(Tg[same] - T[same]).max() / np.abs(T).max()
        print(f"[eigen indef {kind}] stamp {b}: n = {st.n}, lam_min = {lam[0]:.3e}, info = {res.info[b]}, min |lam + kappa| = {gap:.2e}, "
              f"T err {eT:.2e} (tol {tol:.2e}), flips {int((~same).sum())}")
        assert eT <= tol
        assert np.allclose(res.UC[b].cpu().numpy().ravel()[same], UC[same], rtol=1e-5 + tol, atol=1e-9)
        assert np.allclose(res.Sigma[b].cpu().numpy().ravel()[same], Sigma[same], rtol=1e-5 + tol, atol=1e-9)
    assert flips <= (6 if kind == "multi" else 0)


@pytest.mark.parametrize("split", ["1", "2", "4cus"])
@pytest.mark.parametrize("tag", ["eig1", "eigm"])
def test_eigen_batch_of_indefinite_stamps_in_rounds(golden, tag, split, monkeypatch):
    """imcom_solve_eigen on a batch of five stamps -- three indefinite copies of the chain golden's system (N = 220), the
    positive definite original, and an empty stamp -- with the eigenbasis route limited to two stamps at a time
    (IMCOM_EIGEN_FALLBACK_CAP: the route normally takes as many as the workspace holds): info = [1, 0, 1, 0, 1], the indefinite ones
    equal the reference's outputs, the others are untouched by their neighbours' fallback."""
    import ctypes as C

    from pyimcom_amd._lib import MEM_HOST, check, default_context, lib
    from tests.test_oracle import indef_case

    monkeypatch.setenv("IMCOM_EIGEN_FALLBACK_CAP", "2")
    # "2": the batch as two sub-batches on streams of their own, each with indefinite stamps; "4cus": four, every stream confined to a
    # quarter of the CUs (IMCOM_SPLIT_CUS, hipExtStreamCreateWithCUMask: the opt-in of profiles/r04_negative_results.txt item 5)
    monkeypatch.setenv("IMCOM_EIGEN_SPLIT", split.replace("cus", ""))
    if split.endswith("cus"):
        monkeypatch.setenv("IMCOM_SPLIT_CUS", "1")
    g, A, mB, Cs, n2f = indef_case(golden, "chain")
    ch = golden("stamp_chain_mid")
    kC = np.ascontiguousarray(g[f"chain_{tag}_kappaC"], dtype=np.float64)
    n, m, ldn = A.shape[0], mB.shape[1], 256
    batch = 5
    ns = np.array([n, n, n, 0, n], dtype=np.int32)
    Ab = np.zeros((batch, ldn, ldn))
    Bb = np.zeros((batch, m, ldn))
    for s in (0, 2, 4):
        Ab[s, :n, :n], Bb[s, :, :n] = A, mB[0]
    Ab[1, :n, :n], Bb[1, :, :n] = ch["A"], mB[0]
    Cv = np.full(batch, float(Cs[0]))
    T = np.full((batch, m, ldn), np.nan, dtype=np.float32)
    UC, Sg, kp = (np.zeros((batch, m), dtype=np.float32) for _ in range(3))
    info = np.full(batch, -1, dtype=np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    check(lib.imcom_solve_eigen(default_context().handle, batch, p(ns), ldn, m, p(Ab), p(Bb), p(Cv), p(kC), len(kC), float(g["chain_uctarget"]),
                                float(g["chain_sigmamax"]), 13, p(T), p(UC), p(Sg), p(kp), p(info), MEM_HOST))
    assert info.tolist() == [1, 0, 1, 0, 1]
    ref = {q: g[f"chain_{tag}_{q}"] for q in ("T", "UC", "Sigma", "kappa")}
    for s in (0, 2, 4):
        assert np.abs(T[s, :, :n] - ref["T"][0]).max() <= 2e-6 * np.abs(ref["T"]).max(), s
        assert np.all(T[s, :, n:] == 0)
        assert np.allclose(UC[s], ref["UC"].ravel(), rtol=2e-5, atol=1e-9) and np.allclose(Sg[s], ref["Sigma"].ravel(), rtol=2e-5, atol=1e-9)
        assert np.allclose(kp[s], ref["kappa"].ravel(), rtol=1e-6, atol=0)
    # the positive definite stamp: the chain golden's own Eigen outputs for these nodes do not exist; compare with the oracle
    from oracle import oracle as orc

    To, Uo, So, ko, _ = orc.eigen_kernel(ch["A"], mB[0], float(Cs[0]), kC, float(g["chain_uctarget"]), float(g["chain_sigmamax"]))
    same = np.abs(kp[1].astype(np.float64) / ko.astype(np.float64) - 1.0) <= 1e-6
    assert same.sum() >= same.size - 2
    assert np.abs(T[1, :, :n][same] - To[same]).max() <= 2e-6 * np.abs(To).max()
    assert np.all(UC[3] == 1) and np.all(Sg[3] == 0) and np.all(kp[3] == 1)  # lakernel.py:110-119
