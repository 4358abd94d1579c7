"""The Iterative kernel in the regime the reference's users run it: configs/default_config.json -- LAKERNEL
Iterative, KAPPAC [0.0], OUTSIZE [., 32, 0.0390625], INPAD 0.6, ITERRTOL 1.5e-3, ITERMAX 30, six exposures (synth.CONFIGS["iter_default"]),
lakernel.py:397-442 (conjugate_gradient) and 533-654 (IterKernel), the clamp of coadd.py:1104-1107.

At kappa = 0 the sub-systems are singular to rounding (eigenvalues of A down to -3e-12 against 0.085), the recurrences stop between
13 and 30 steps, and a recurrence's iterate after k steps carries the rounding of every inner product amplified along the run: numpy
with another order of its sums differs from itself (shown below, on the oracle alone).  Parity therefore is stated the way that
survives this: the acceptance discs exactly; the same number of steps for nearly all output pixels, and where the steps agree T to a
few 1e-4 of its largest entry (median 1e-7: float32's own rounding); where the stopping test fell the other way -- one sum being
a rounding above the tolerance, the other below -- BOTH sides return an iterate that meets the reference's stopping rule (residual below
rtol |b|, or ITERMAX steps used); and the reference's own image criterion, Iterative against Cholesky < 2.5e-3 rms
(tests/pyimcom/test_pyimcom.py:971-978)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_inputs(cfg, psfs, target):
    from oracle import oracle as orc

    g, tabs_ref, C_ref = orc.stamp_tables(cfg, psfs, target)
    E = cfg.n_expo
    tri = lambda i, j: (2 * E - i + 1) * i // 2 + j - i  # noqa: E731
    tab = np.array([[tri(a, b) if a <= b else (tri(b, a) | (1 << 30)) for b in range(E)] for a in range(E)], dtype=np.int32)
    pen = np.array([[-cfg.flat_penalty / E + (cfg.flat_penalty if a == b else 0.0) for b in range(E)] for a in range(E)])
    return g, tabs_ref, float(C_ref[0]), tab, pen, np.arange(E) + E * (E + 1) // 2


def test_iter_default_full_size_vs_oracle():
    import torch

    from oracle import oracle as orc
    from pyimcom_amd import synth
    from pyimcom_amd._lib import default_context
    from pyimcom_amd.stamps import PSFGroupTables, StampBatch
    from tests.parity import iter_parity

    cfg = synth.CONFIGS["iter_default"]
    ctx = default_context(0)
    stamps = [synth.make_stamp(cfg, i) for i in range(2)]
    psfs, target = synth.make_psfs(cfg, cfg.n_expo)
    tables = PSFGroupTables(psfs, target, cfg.nfft, ctx=ctx)
    b = StampBatch(cfg, stamps, tables, ctx=ctx)
    b.run()
    torch.cuda.synchronize()
    stats, steps_all = ctx.iter_stats(2 * cfg.m)
    assert stats["blocked"] and 600 < stats["max_union"] <= 1024 and stats["patches"] == 2 * 64  # the blocked solver, not the per-pixel fallback
    res = b.result()

    g, tabs_ref, C, tab, pen, io = _oracle_inputs(cfg, psfs, target)
    st = stamps[0]
    ref = orc.stamp_full(cfg, g, tabs_ref, C, st, tab, pen, io)
    g1 = np.arange(cfg.n2f, dtype=np.float64)
    oy, ox = np.repeat(st.out_y0 + g1, cfg.n2f), np.tile(st.out_x0 + g1, cfg.n2f)
    mB = np.ascontiguousarray(ref["Bt"].T)
    osteps = []
    Tr = orc.iter_kernel(ref["A"], mB, C, np.array(cfg.kappaC), cfg.uctarget, cfg.sigmamax, oy, ox, st.y, st.x, cfg.rho, cfg.iter_rtol, cfg.iter_max, steps=osteps)[0]
    osteps = np.array(osteps)
    assert np.array_equal(Tr, ref["T"])  # (stamp_full ran the same kernel)
    Tg = res.T(0).cpu().numpy()
    relevant = orc._relevant(oy, ox, st.y, st.x, cfg.rho)

    assert 540 < relevant.sum(axis=1).mean() < 580  # ~560 input pixels per output pixel: the regime this test is about
    # acceptance discs, steps, T, the stopping rule on both sides, the maps: tests/parity.py iter_parity (its docstring is the statement)
    gsteps = steps_all[: cfg.m]
    UCr, Sr = ref["UC"].ravel(), ref["Sigma"].ravel()
    UCg, Sg = res.UC[0].cpu().numpy().ravel(), res.Sigma[0].cpu().numpy().ravel()
    rep = iter_parity(ref["A"], mB, C, relevant, cfg.iter_rtol, cfg.iter_max, (Tg, gsteps, UCg, Sg), (Tr, osteps, UCr, Sr))
    assert 13 <= osteps.min() and osteps.max() <= cfg.iter_max and (UCg >= 1e-32).all() and (Sg >= 1e-32).all()  # (the clamp of coadd.py:1104-1107)
    same = gsteps == osteps
    assert np.array_equal(res.kappa[0].cpu().numpy(), ref["kappa"])
    img_g, img_r = res.outimage[0].cpu().numpy().reshape(cfg.n_inframe, -1), ref["outimage"].reshape(cfg.n_inframe, -1)
    bound = np.abs(Tr).astype(np.float64) @ np.abs(st.indata.astype(np.float64)).T  # sum_i |T_ai| |indata_i|, [m][n_inframe]
    moved = np.abs(Tg.astype(np.float64) - Tr) @ np.abs(st.indata.astype(np.float64)).T  # what the difference of the two T can move a coadded pixel by
    assert (np.abs(img_g - img_r) <= (moved + 2e-5 * bound).T + 1e-7).all()  # (2e-5: float32 accumulation, tests/parity.py TOL["image"])

    # what "the reference against itself" looks like: tests/test_oracle.py::test_iter_default_oracle_against_itself runs the oracle's CG on
    # the same stamp with every selection in reverse order (the same recurrences, other sums): 1.7 % of the pixels stop a step apart, T
    # where the steps agree within 5.3e-4 -- the device's run differs from the oracle's by as much and no more
    assert (~same).mean() < 0.05 and rep["dT_same_max"] < 1.5e-3, rep


def test_iterative_against_cholesky_image_on_a_block():
    """tests/pyimcom/test_pyimcom.py:971-978 on a synthetic block: the science layer coadded by the Iterative kernel (default
    configuration: kappa = 0, 30 steps) against the Cholesky kernel's at the kappa/C = 5e-4 of that test: std < 2.5e-3, |mean| < 2e-4."""
    import dataclasses

    import torch

    from pyimcom_amd import synth
    from pyimcom_amd.blockrun import coadd_block
    from pyimcom_amd.select import InStampPool
    from pyimcom_amd.stamps import PSFGroupTables

    cfg = synth.CONFIGS["iter_default"]
    n1P, E = 4, cfg.n_expo
    rng = np.random.default_rng(11)
    inst = synth.make_instamps(cfg, n1P, E, rng)
    p = synth.NATIVE_ARCSEC / cfg.dtheta_as
    sx, sy = rng.uniform(0, n1P * cfg.n2, 6), rng.uniform(0, n1P * cfg.n2, 6)
    inst2 = []
    for px, py, data, cum in inst:  # science layer: unit-flux stars through a Gaussian of 0.9 native pixels
        d = data.copy()
        d[0] = (np.exp(-0.5 * ((px[:, None] - sx[None]) ** 2 + (py[:, None] - sy[None]) ** 2) / (0.9 * p) ** 2).sum(axis=1) / (2 * np.pi * 0.9**2)).astype(np.float32)
        inst2.append((px, py, d, cum))
    pool = InStampPool(inst2, cfg.n_inframe)
    psfs, target = synth.make_psfs(cfg, E)
    tabs = PSFGroupTables(psfs, target, cfg.nfft)
    it = coadd_block(cfg, pool, tabs, n1P, E, pad_sides=None)
    ch = coadd_block(dataclasses.replace(cfg, kernel="Cholesky", kappaC=(5e-4,)), pool, tabs, n1P, E, pad_sides=None)
    torch.cuda.synchronize()
    a, c = it.out_map[0, 0].cpu().numpy(), ch.out_map[0, 0].cpu().numpy()
    lo, hi = cfg.n2 // 2, n1P * cfg.n2 - cfg.n2 // 2
    d = (a - c)[lo:hi, lo:hi]
    assert np.abs(c).max() > 0.05 and d.std() < 2.5e-3 and abs(d.mean()) < 2e-4, (d.std(), d.mean(), np.abs(c).max())
    # the Iterative maps are clamped (coadd.py:1104-1107) and finite
    for k in ("UC", "Sigma"):
        m_ = it.maps[k].cpu().numpy()
        assert np.isfinite(m_).all() and (m_[..., lo:hi, lo:hi] >= 1e-32).all()


def test_full_storage_kernel_agrees_with_the_half_storage_default():
    """The blocked CG on the tiles on and below the diagonal only (csrc/iter_block.hip iter_block_cg_sym_kernel: every tile used for
    q_I += T P_K and q_K += T^T P_I; half the bytes, a fixed order of sums) is the default for unions up to 768 rows; IMCOM_ITER_SYM=0 selects
    the full-storage kernel (which also serves the larger unions).  In a subprocess, one default-configuration stamp through the
    full-storage kernel against the default of this process: the same steps for nearly every pixel, T where they agree within 2e-3 of
    its largest entry."""
    import os
    import subprocess
    import sys

    import torch

    from pyimcom_amd import synth
    from pyimcom_amd._lib import default_context
    from pyimcom_amd.stamps import PSFGroupTables, StampBatch

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, torch, sys\n"
            "from pyimcom_amd import synth\nfrom pyimcom_amd._lib import default_context\nfrom pyimcom_amd.stamps import PSFGroupTables, StampBatch\n"
            "cfg = synth.CONFIGS['iter_default']; ctx = default_context(0)\n"
            "psfs, target = synth.make_psfs(cfg, cfg.n_expo)\n"
            "b = StampBatch(cfg, [synth.make_stamp(cfg, 0)], PSFGroupTables(psfs, target, cfg.nfft, ctx=ctx), ctx=ctx); b.run(); torch.cuda.synchronize()\n"
            "st, steps = ctx.iter_stats(cfg.m)\n"
            "assert not st['half_storage'] and st['bytes'] == st['bytes_full_storage'], st\n"
            "np.savez(sys.argv[1], T=b.result().T(0).cpu().numpy(), steps=steps)\nprint('full ok')\n")
    out = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"imcom_full_{os.getpid()}.npz")
    p = subprocess.run([sys.executable, "-c", code, out], cwd=root, env=dict(os.environ, PYTHONPATH=root, IMCOM_ITER_SYM="0"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "full ok" in p.stdout, p.stdout[-1000:] + p.stderr[-2000:]
    z = np.load(out)
    os.remove(out)
    cfg = synth.CONFIGS["iter_default"]
    ctx = default_context(0)
    psfs, target = synth.make_psfs(cfg, cfg.n_expo)
    b = StampBatch(cfg, [synth.make_stamp(cfg, 0)], PSFGroupTables(psfs, target, cfg.nfft, ctx=ctx), ctx=ctx)
    b.run()
    torch.cuda.synchronize()
    st, steps = ctx.iter_stats(cfg.m)
    assert st["half_storage"] and st["bytes"] < 0.52 * st["bytes_full_storage"]
    T = b.result().T(0).cpu().numpy()
    same = steps == z["steps"]
    d = np.abs(T - z["T"]).max(axis=1) / np.abs(T).max(axis=1)
    assert same.mean() > 0.95 and d[same].max() < 2e-3 and np.median(d) < 1e-6, (same.mean(), d[same].max(), np.median(d))


@pytest.mark.parametrize("n_expo, blocked, inpad", [(8, True, 0.6), (10, False, 0.6), (6, True, 0.61), (7, True, 0.6)])
def test_deeper_stacks_take_the_wide_kernel_or_the_per_pixel_one(n_expo, blocked, inpad):
    """The default configuration at other exposure depths.  Eight exposures: ~750 input pixels per acceptance disc, a 4 x 4 patch's union
    ~950 rows -- the full-storage solver's widest variant (up to 1024 rows).  Seven: 828 rows, the half-storage kernel's largest.  Ten: the unions pass 1024 (1190) and the call falls back to the
    per-pixel kernel (one workgroup per output pixel, lakernel.py:545-586 as written).  Six exposures with INPAD 0.61": unions of 740 rows
    = 47 tiles -- the half-storage kernel where a wave's last tile row ends before the last column panel (4 x 12 - 3 = 45 < 47: the
    panels beyond it are still ended by every wave).  All against the oracle on one stamp with the parity statement of
    tests/parity.py iter_parity."""
    import dataclasses

    import torch

    from oracle import oracle as orc
    from pyimcom_amd import synth
    from pyimcom_amd._lib import default_context
    from pyimcom_amd.stamps import PSFGroupTables, StampBatch
    from tests.parity import iter_parity

    # (16 x 16 outputs instead of 32 x 32: the discs and unions are the same size -- they depend on rho and the depth -- and the oracle's 256
    # recurrences take a quarter of the time)
    cfg = dataclasses.replace(synth.CONFIGS["iter_default"], n2=16, n_expo=n_expo, inpad_as=inpad, name=f"iter_default_e{n_expo}",
                              iter_max=30 if blocked else 8)  # (the per-pixel fallback: eight steps are enough to test it, and the oracle's 1400-row recurrences are slow)
    ctx = default_context(0)
    st = synth.make_stamp(cfg, 0)
    psfs, target = synth.make_psfs(cfg, n_expo)
    b = StampBatch(cfg, [st], PSFGroupTables(psfs, target, cfg.nfft, ctx=ctx), ctx=ctx)
    b.run()
    torch.cuda.synchronize()
    stats, gsteps = ctx.iter_stats(cfg.m)
    if inpad > 0.6:
        assert stats["blocked"] and stats["half_storage"] and 720 < stats["max_union"] <= 768, stats
    elif n_expo == 7:  # 828 rows = 52 tiles: the half-storage kernel's seven-rows-per-wave variant (its LDS limit: 864 rows)
        assert stats["blocked"] and stats["half_storage"] and 768 < stats["max_union"] <= 864, stats
    else:
        assert stats["blocked"] == blocked and not stats["half_storage"] and (864 < stats["max_union"] <= 1024 if blocked else stats["max_union"] > 1024), stats
    res = b.result()
    g, tabs_ref, C, tab, pen, io = _oracle_inputs(cfg, psfs, target)
    A, Bt = orc.stamp_system(g, st.x, st.y, st.expo, tabs_ref, tab, pen, io, st.out_x0, st.out_y0, cfg.n2f)
    mB = np.ascontiguousarray(Bt.T)
    g1 = np.arange(cfg.n2f, dtype=np.float64)
    oy, ox = np.repeat(st.out_y0 + g1, cfg.n2f), np.tile(st.out_x0 + g1, cfg.n2f)
    osteps = []
    Tr, UCr, Sr, _, _ = orc.iter_kernel(A, mB, C, np.array(cfg.kappaC), cfg.uctarget, cfg.sigmamax, oy, ox, st.y, st.x, cfg.rho, cfg.iter_rtol, cfg.iter_max, steps=osteps)
    UCr, Sr = orc.iterative_clamp(UCr, Sr)
    relevant = orc._relevant(oy, ox, st.y, st.x, cfg.rho)
    assert relevant.sum(axis=1).mean() > 90 * n_expo
    rep = iter_parity(A, mB, C, relevant, cfg.iter_rtol, cfg.iter_max, (res.T(0).cpu().numpy(), gsteps, res.UC[0].cpu().numpy().ravel(), res.Sigma[0].cpu().numpy().ravel()),
                      (Tr, np.array(osteps), UCr, Sr), min_same=0.93)
    assert rep["dT_same_max"] < 2e-3 and (blocked or (gsteps == 8).all()), rep


@pytest.mark.parametrize("kernel", ["Cholesky", "Iterative"])
def test_point_source_known_answer_on_a_block(kernel):
    """The reference's end-to-end known answer (tests/pyimcom/test_pyimcom.py:938-951) on a synthetic block: unit-flux stars observed through
    every exposure's own PSF must come out of the coaddition as the TARGET PSF at the stars' positions with amplitude 1 --
    SL1 = sum(p d) / sum(p^2) within 5e-4 of 1 and VAR = sum((d - SL1 p)^2) / sum(p^2) < 1e-5, p the target Gaussian around the stars
    (in units of surface brightness per native pixel).  Exercises the whole device chain: tables from the PSFs, selection, A, B, the LA
    kernel, the coaddition, the block maps.  cfg-1 geometry (Gaussian PSFs of 0.9 + 0.05 e native pixels, 32 x 32 outputs, INPAD 0.6")."""
    import dataclasses

    import torch

    from pyimcom_amd import synth
    from pyimcom_amd.blockrun import coadd_block
    from pyimcom_amd.select import InStampPool
    from pyimcom_amd.stamps import PSFGroupTables

    base = synth.CONFIGS["cfg1"]
    cfg = dataclasses.replace(base, kernel=kernel, kappaC=(5e-4,) if kernel == "Cholesky" else (0.0,), n_inframe=1, name=f"cfg1_{kernel}")
    n1P, E = 4, cfg.n_expo
    rng = np.random.default_rng(3)
    inst = synth.make_instamps(cfg, n1P, E, rng)
    p = synth.NATIVE_ARCSEC / cfg.dtheta_as  # output pixels per native pixel
    span = n1P * cfg.n2
    sx, sy = rng.uniform(0.3 * span, 0.7 * span, 3), rng.uniform(0.3 * span, 0.7 * span, 3)
    inst2 = []
    for px, py, data, cum in inst:
        d = np.zeros((1, px.size), np.float32)
        for e in range(E):  # exposure-major inside a cell (make_instamps): pixels cum[e] .. cum[e + 1] belong to exposure e
            sl = slice(int(cum[e]), int(cum[e + 1]))
            sig = (cfg.psf_sigma + 0.05 * e) * p  # the exposure's PSF (synth.make_psfs) in output pixels
            r2 = (px[sl, None] - sx[None]) ** 2 + (py[sl, None] - sy[None]) ** 2
            d[0, sl] = (np.exp(-0.5 * r2 / sig**2) / (2 * np.pi * (sig / p) ** 2)).sum(axis=1)
        inst2.append((px, py, d, cum))
    pool = InStampPool(inst2, 1)
    psfs, target = synth.make_psfs(cfg, E)
    tabs = PSFGroupTables(psfs, target, cfg.nfft)
    maps = coadd_block(cfg, pool, tabs, n1P, E, pad_sides=None)
    torch.cuda.synchronize()
    dmap = maps.out_map[0, 0].cpu().numpy().astype(np.float64)
    yy, xx = np.mgrid[0:span, 0:span].astype(np.float64)
    sig_t = cfg.extrasmooth * p
    pred = sum(np.exp(-0.5 * ((xx - x0) ** 2 + (yy - y0) ** 2) / sig_t**2) for x0, y0 in zip(sx, sy)) / (2 * np.pi * cfg.extrasmooth**2)
    lo, hi = cfg.n2 // 2, span - cfg.n2 // 2
    pw, dw = pred[lo:hi, lo:hi], dmap[lo:hi, lo:hi]
    SL1 = float((pw * dw).sum() / (pw**2).sum())
    VAR = float(((dw - SL1 * pw) ** 2).sum() / (pw**2).sum())
    uc = float(maps.maps["UC"][0].cpu().numpy()[lo:hi, lo:hi].mean())
    print(f"[point source] {kernel}: SL1 - 1 = {SL1 - 1:.3e}, VAR = {VAR:.3e}, mean U/C = {uc:.3e}")
    assert abs(SL1 - 1) < (5e-4 if kernel == "Cholesky" else 1e-3) and 0.5 * uc < VAR < 2 * uc and VAR < (1e-4 if kernel == "Cholesky" else 1e-3), (kernel, SL1, VAR, uc)
