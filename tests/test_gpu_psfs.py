"""GPU parity of PSF sampling and the analytic target PSFs against the reference's own outputs
(tests/golden/make_golden_psf.py; psfutil.py:117-223, 709-795, 615-671)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_target_psf_images(golden):
    from pyimcom_amd import psfs

    g = golden("psfovl")
    a = psfs.psf_gaussian(16, 2.5, 2.5)
    assert np.allclose(a, g["gauss_16_25"], rtol=1e-14, atol=0)
    b = psfs.psf_simple_airy(24, 5.0, obsc=0.31, tophat_conv=4.0, sigma=1.2)
    assert np.abs(b - g["airy_24"]).max() <= 2e-13 * np.abs(g["airy_24"]).max()


def test_sample_psf_golden(golden):
    import torch

    from pyimcom_amd import psfs

    g = golden("psf_sample")
    ns = int(g["nsamp"])
    a = psfs.sample_psf(g["in_psf"][None], ns, g["in_yxco"][None])[0]
    assert np.abs(a - g["in_psf_arr"]).max() <= 1e-13 * np.abs(g["in_psf_arr"]).max()
    b = psfs.sample_psf(g["grid_psf"], ns)
    assert np.abs(b - g["grid_psf_arr"]).max() <= 1e-13 * np.abs(g["grid_psf_arr"]).max()
    # device in / device out gives the same numbers
    bt = psfs.sample_psf(torch.as_tensor(g["grid_psf"], device="cuda:0"), ns)
    assert np.array_equal(bt.cpu().numpy(), b)


@pytest.mark.parametrize("name,kind", [("gauss", "GAUSSIAN"), ("airyobsc", "AIRYOBSC"), ("airyunobsc", "AIRYUNOBSC")])
def test_output_psf_group_golden(golden, name, kind):
    """PSFGrp(in_or_out=False): _get_outpsf -> _sample_psf(None, .) -> circular cut-out -> normalisation."""
    from pyimcom_amd import psfs

    g = golden("psf_sample")
    ns, ov = int(g["nsamp"]), int(g["oversamp"])
    sig, uf, circ, norm = g[f"out_{name}_pars"]
    img = psfs.get_outpsf(kind, float(sig), int(uf), ns, ov)
    arr = psfs.sample_psf(img[None], ns, None, bool(circ), bool(norm))[0]
    ref = g[f"out_{name}"]
    assert np.abs(arr - ref).max() <= 3e-13 * np.abs(ref).max()
    assert np.array_equal(arr == 0, ref == 0)  # the cut-out mask


def test_group_tables_from_images(golden):
    """Raw PSF images + sampling positions -> sampled PSFs -> overlap tables, all on the device, equals the
    oracle's chain (sample_psf, finish_psf_group, pad_and_rfft2, overlap_*)."""
    from oracle import oracle as orc
    from pyimcom_amd.stamps import PSFGroupTables

    g = golden("psf_sample")
    ns, ov = int(g["nsamp"]), int(g["oversamp"])
    nfft = 2 * (ns + 1)
    rng = np.random.default_rng(2)
    imgs = np.stack([g["in_psf"], g["in_psf"][::-1, ::-1] * 0.9 + 0.01 * rng.standard_normal(g["in_psf"].shape)])
    yxco = np.stack([g["in_yxco"], 1.02 * g["in_yxco"][:, ::-1]])
    tabs = PSFGroupTables.from_images(imgs, yxco, ("GAUSSIAN", 1.2, 2), ns, nfft, ov, psf_circ=True, psf_norm=True)
    pin = np.stack([orc.sample_psf(imgs[e], ns, yxco[e]) for e in range(2)])
    orc.finish_psf_group(pin, True, True)
    pout = orc.sample_psf(orc.get_outpsf("GAUSSIAN", 1.2, 2, ns, ov), ns)[None].copy()
    orc.finish_psf_group(pout, True, True)
    geo = orc.Geom(int(g["npixpsf"]), ov, float(g["dtheta_as"]) / 3600.0, 0.0)
    r_in, r_out = orc.pad_and_rfft2(pin, geo), orc.pad_and_rfft2(pout, geo)
    ref = np.concatenate([orc.overlap_self(r_in, geo), orc.overlap_cross(r_in, r_out, geo)[:, 0]])
    got = tabs.tables.cpu().numpy()[:, 6:-6, 6:-6]
    assert np.abs(got - ref).max() <= 5e-13 * np.abs(ref).max()
    assert abs(tabs.C - orc.overlap_out_C(r_out, geo)[0]) <= 1e-12 * tabs.C


def test_amp_penalty_tables_golden(golden):
    """cfg.amp_penalty (psfutil.py:661-671): the reweighted spectra through to the overlap table and C."""
    from pyimcom_amd import psfs
    from pyimcom_amd.stamps import PSFGroupTables

    g = golden("psf_sample")
    ns, ov = int(g["nsamp"]), int(g["oversamp"])
    a0, a1 = g["amp_penalty"]
    mk = lambda kind, sig: psfs.sample_psf(psfs.get_outpsf(kind, sig, 2, ns, ov)[None], ns, None, True, True)  # noqa: E731
    tabs = PSFGroupTables(mk("GAUSSIAN", 1.2), mk("AIRYOBSC", 0.9), 2 * (ns + 1), amp_penalty=(a0, a1 * ov))
    got = tabs.tables[tabs.ntri].cpu().numpy()[6:-6, 6:-6]
    ref = g["amp_ovl_io"][0, 0]
    assert np.abs(got - ref).max() <= 5e-13 * np.abs(ref).max()
    assert abs(tabs.C - g["amp_outovlc"][0]) <= 1e-12 * tabs.C
    plain = PSFGroupTables(mk("GAUSSIAN", 1.2), mk("AIRYOBSC", 0.9), 2 * (ns + 1))
    assert abs(plain.C - tabs.C) > 0.05 * tabs.C  # the weighting really changes the numbers


def test_airy_known_answers_of_the_reference():
    """tests/pyimcom/test_psf.py:20-60 (centre value, sum, FWHM-free part) and 71-82 (output PSF peak ratios)."""
    from pyimcom_amd import psfs

    im = psfs.psf_simple_airy(100, 4.0)
    assert abs(im[50, 50] - 0.045421877940855226) < 0.001 and abs(im.sum() - 0.9853733474017817) < 0.001
    im = psfs.psf_simple_airy(100, 4.0, obsc=0.5)
    assert abs(im[50, 50] - 0.03339794632862726) < 0.001 and abs(im.sum() - 0.970256598273068) < 0.001
    ns, ov = 25 * 4 - 1, 4
    mg = psfs.get_outpsf("GAUSSIAN", 0.3, 2, ns, ov).max()
    mau = psfs.get_outpsf("AIRYUNOBSC", 0.3, 2, ns, ov).max()
    mao = psfs.get_outpsf("AIRYOBSC", 0.3, 2, ns, ov).max()
    assert 0.2 < mau / mg < 0.3 and 0.8 < mao / mau < 0.9


@pytest.mark.parametrize("npixpsf,oversamp", [(5, 6), (10, 4), (9, 2), (7, 4), (3, 4), (4, 4), (30, 16), (32, 8), (24, 16), (32, 16)])
def test_psf_overlap_mixed_radix_vs_oracle(npixpsf, oversamp):
    """Table geometries whose nfft exercises every radix of the general line-FFT kernels (60 = 4 3 5, 80 = 16 5, 36 = 4 3 3,
    24 = 8 3, 32 = 16 2, 960 = 16 4 3 5), one that has none of them (56 = 8 7: the dense-DFT fallback), and the three static
    16 x 16 x r shapes of the persistent kernels (512, 768, 1024), all against the oracle's numpy FFTs, with and without the
    amp_penalty weighting."""
    import ctypes as C

    import torch

    from oracle import oracle as orc
    from pyimcom_amd._lib import check, default_context, lib

    geo = orc.Geom(npixpsf, oversamp, 0.04 / 3600.0, 0.0)
    ns, nfft = geo.nsamp, geo.nfft
    rng = np.random.default_rng(ns)
    yy, xx = np.mgrid[:ns, :ns] - ns // 2
    p1 = np.stack([np.exp(-(xx**2 + yy**2) / (2.0 * (2.0 + 0.3 * k) ** 2)) + 0.01 * rng.standard_normal((ns, ns)) for k in range(3)])
    p2 = np.stack([np.exp(-((xx - 0.7) ** 2 + (yy + 0.4) ** 2) / (2.0 * (2.5 + 0.2 * k) ** 2)) for k in range(2)])
    amp = np.array([0.3, 0.45 * oversamp])
    r1, r2 = orc.pad_and_rfft2(p1, geo), orc.pad_and_rfft2(p2, geo)
    # Fourier-mode reweighting of PSFGrp.__init__ (psfutil.py:661-671): 1 + a0 exp(-2 pi^2 |u|^2 a1^2), u in cycles per sample
    uy, ux = np.fft.fftfreq(nfft)[:, None], np.fft.rfftfreq(nfft)[None, :]
    w = 1.0 + amp[0] * np.exp(-2.0 * np.pi**2 * (ux**2 + uy**2) * amp[1] ** 2)
    dev = torch.device("cuda:0")
    ctx = default_context()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    t1, t2 = torch.as_tensor(p1, device=dev), torch.as_tensor(p2, device=dev)
    dp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    for use_amp in (False, True):
        a1, a2 = (r1 * w, r2 * w) if use_amp else (r1, r2)
        ref = orc.overlap_cross(a1, a2, geo)
        pairs = np.array([(i, j) for i in range(3) for j in range(2)], dtype=np.int32)
        out = torch.empty((len(pairs), ns + 12, ns + 12), dtype=torch.float64, device=dev)
        check(lib.imcom_psf_overlap(ctx.handle, dp(t1), 3, dp(t2), 2, ns, nfft, pairs.ctypes.data_as(C.c_void_p), len(pairs),
                                    amp.ctypes.data_as(C.c_void_p) if use_amp else None, dp(out)))
        got = out.cpu().numpy()[:, 6:-6, 6:-6].reshape(3, 2, ns, ns)
        assert np.abs(got - ref).max() < 2e-13 * np.abs(ref).max(), (nfft, use_amp, np.abs(got - ref).max() / np.abs(ref).max())


def test_smooth_and_pad_golden(golden):
    """imcom_smooth_and_pad (circulant products on the GEMM engine) vs the reference function's own outputs; host and
    device buffers, single image and stack."""
    import torch

    from pyimcom_amd import psfs

    g = golden("smooth_pad")
    for name in "abcde":
        w, sg = (float(v) for v in g[f"{name}_pars"])
        ref = g[f"{name}_out"]
        got = psfs.smooth_and_pad(g[f"{name}_in"], w, sg)
        assert isinstance(got, np.ndarray) and got.shape == ref.shape
        assert np.abs(got - ref).max() <= 2e-14 * np.abs(ref).max(), (name, np.abs(got - ref).max())
    # a stack on the device: three copies of one image come out identical to the single call
    img = torch.as_tensor(g["b_in"], device="cuda:0")
    w, sg = (float(v) for v in g["b_pars"])
    st = psfs.smooth_and_pad(torch.stack([img, 2.0 * img, img]), w, sg)
    assert st.shape == (3,) + g["b_out"].shape
    one = psfs.smooth_and_pad(img, w, sg)
    assert torch.equal(st[0], one) and torch.equal(st[2], one) and torch.allclose(st[1], 2.0 * one, rtol=1e-15, atol=0)
    assert np.abs(one.cpu().numpy() - g["b_out"]).max() <= 2e-14 * np.abs(g["b_out"]).max()


@pytest.mark.parametrize("ns,nfft", [(201, 512), (301, 768), (255, 1024), (33, 80), (21, 60)])
def test_psf_overlap_padding_beyond_twice_nsamp(ns, nfft):
    """nfft only has to be >= 2 nsamp: tables with more zero padding than PSFGrp.setup uses (static and general line-FFT
    kernels: the kept window is then a smaller part of the transform) against numpy's rfft2 / irfft2, borders exactly zero."""
    import ctypes as C

    import torch

    from pyimcom_amd._lib import check, default_context, lib

    rng = np.random.default_rng(ns + nfft)
    yy, xx = np.mgrid[:ns, :ns] - ns // 2
    p1 = np.stack([np.exp(-(xx**2 + yy**2) / (2.0 * (2.0 + 0.4 * k) ** 2)) + 0.01 * rng.standard_normal((ns, ns)) for k in range(3)])
    p2 = np.stack([np.exp(-((xx - 0.5) ** 2 + (yy + 0.3) ** 2) / (2.0 * (2.3 + 0.3 * k) ** 2)) for k in range(2)])
    f1, f2 = np.zeros((3, nfft, nfft)), np.zeros((2, nfft, nfft))
    f1[:, :ns, :ns], f2[:, :ns, :ns] = p1, p2
    r1, r2 = np.fft.rfft2(f1), np.fft.rfft2(f2)
    pairs = np.array([(i, j) for i in range(3) for j in range(2)], dtype=np.int32)
    ref = np.stack([np.roll(np.fft.irfft2(r1[i] * np.conj(r2[j]), s=(nfft, nfft)), (ns // 2, ns // 2), axis=(0, 1))[:ns, :ns] for i, j in pairs])
    dev = torch.device("cuda:0")
    ctx = default_context()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    t1, t2 = torch.as_tensor(p1, device=dev), torch.as_tensor(p2, device=dev)
    out = torch.full((len(pairs), ns + 12, ns + 12), np.nan, dtype=torch.float64, device=dev)
    dp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    check(lib.imcom_psf_overlap(ctx.handle, dp(t1), 3, dp(t2), 2, ns, nfft, pairs.ctypes.data_as(C.c_void_p), len(pairs), None, dp(out)))
    got = out.cpu().numpy()
    assert np.abs(got[:, 6:-6, 6:-6] - ref).max() < 2e-13 * np.abs(ref).max()
    border = got.copy()
    border[:, 6:-6, 6:-6] = 0.0
    assert np.all(border == 0.0)  # every border element written, none left at the NaN fill


@pytest.mark.gpu
@pytest.mark.parametrize("ns,nfft", [(255, 512), (301, 768), (33, 80)])
def test_psf_overlap_windows(ns, nfft):
    """imcom_psf_overlap_spectra_win through the C-ABI: inside a pair's window the table is bit for bit the one of the plain call;
    with the static line-FFT kernels (nfft = 512, 768) nothing outside the window (and the zero border next to it) is written, the
    general kernels (nfft = 80) ignore the windows and fill whole tables; a window out of range is refused."""
    import ctypes as C

    import torch

    from pyimcom_amd._lib import ImcomError, check, default_context, lib

    rng = np.random.default_rng(ns)
    yy, xx = np.mgrid[:ns, :ns] - ns // 2
    p = np.stack([np.exp(-(xx**2 + yy**2) / (2.0 * (2.0 + 0.4 * k) ** 2)) + 0.01 * rng.standard_normal((ns, ns)) for k in range(3)])
    dev = torch.device("cuda:0")
    ctx = default_context()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    t = torch.as_tensor(p, device=dev)
    size = int(lib.imcom_psf_spectra_size(ns, nfft))
    assert size > 0
    spec = torch.empty((3, size), dtype=torch.float64, device=dev)
    dp = lambda a: C.c_void_p(a.data_ptr())  # noqa: E731
    check(lib.imcom_psf_spectra(ctx.handle, dp(t), 3, ns, nfft, dp(spec)))
    pairs = np.array([(0, 1), (1, 2), (2, 0), (1, 1)], dtype=np.int32)
    nc = ns // 2
    win = np.array([[0, nc + 8, 0, ns], [nc - 8, ns, nc - 8, ns], [3, nc, 0, nc + 8], [0, ns, 0, ns]], dtype=np.int32)
    full = torch.empty((4, ns + 12, ns + 12), dtype=torch.float64, device=dev)
    check(lib.imcom_psf_overlap_spectra(ctx.handle, dp(spec), 3, dp(spec), 3, ns, nfft, pairs.ctypes.data_as(C.c_void_p), 4, None, dp(full)))
    part = torch.full((4, ns + 12, ns + 12), -7.0, dtype=torch.float64, device=dev)
    check(lib.imcom_psf_overlap_spectra_win(ctx.handle, dp(spec), 3, dp(spec), 3, ns, nfft, pairs.ctypes.data_as(C.c_void_p), 4, None,
                                            win.ctypes.data_as(C.c_void_p), dp(part)))
    torch.cuda.synchronize()
    f, g = full.cpu().numpy(), part.cpu().numpy()
    static = nfft in (512, 768, 1024)
    for q, (r0, r1, c0, c1) in enumerate(win):
        assert np.array_equal(g[q, 6 + r0 : 6 + r1, 6 + c0 : 6 + c1], f[q, 6 + r0 : 6 + r1, 6 + c0 : 6 + c1]), q
        if static:
            inside = np.zeros(g[q].shape, bool)
            lo, hi = 2 * (r0 // 2), min(2 * ((r1 + 1) // 2), ns)  # the row transform works on row pairs
            inside[6 + lo : 6 + hi, 6 + c0 : 6 + c1] = True
            untouched = g[q] == -7.0
            assert np.all(untouched[~inside & (f[q] != 0.0)]), q  # outside the window only zero-border cells may have been written
            assert not untouched[inside].any(), q
        else:
            assert np.array_equal(g[q], f[q])
    bad = win.copy()
    bad[0, 1] = ns + 1
    with pytest.raises(ImcomError):
        check(lib.imcom_psf_overlap_spectra_win(ctx.handle, dp(spec), 3, dp(spec), 3, ns, nfft, pairs.ctypes.data_as(C.c_void_p), 4, None,
                                                bad.ctypes.data_as(C.c_void_p), dp(part)))
