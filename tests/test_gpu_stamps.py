"""GPU parity of the device-resident stamp path (PSF overlap tables -> A, B -> solve -> coaddition)
against the CPU oracle, through the C-ABI.  Run with -m gpu on an MI355X."""

import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_psf_overlap_golden(golden):
    """imcom_psf_overlap vs the reference's PSFOvl tables (tests/golden/psfovl.npz)."""
    import torch

    from pyimcom_amd._lib import check, default_context, lib

    g = golden("psfovl")
    ns, nfft = int(g["nsamp"]), int(g["nfft"])
    dev = torch.device("cuda:0")
    ctx = default_context()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    p1, p2, po = (torch.as_tensor(np.ascontiguousarray(g[k]), device=dev) for k in ("psf1", "psf2", "psfo"))
    dp = lambda t: C.c_void_p(t.data_ptr())

    def run(a, b, pairs):
        pairs = np.array(pairs, dtype=np.int32)
        out = torch.empty((len(pairs), ns + 12, ns + 12), dtype=torch.float64, device=dev)
        check(lib.imcom_psf_overlap(ctx.handle, dp(a), a.shape[0], dp(b), b.shape[0], ns, nfft,
                                    pairs.ctypes.data_as(C.c_void_p), len(pairs), None, dp(out)))
        o = out.cpu().numpy()
        assert np.all(o[:, :6] == 0) and np.all(o[:, -6:] == 0) and np.all(o[:, :, :6] == 0) and np.all(o[:, :, -6:] == 0)
        return o[:, 6:-6, 6:-6]

    tri = run(p1, p1, [(i, j) for i in range(3) for j in range(i, 3)])
    assert np.abs(tri - g["ovl_self"]).max() < 2e-13 * np.abs(g["ovl_self"]).max()
    cross = run(p1, p2, [(i, j) for i in range(3) for j in range(3)]).reshape(3, 3, ns, ns)
    assert np.abs(cross - g["ovl_cross"]).max() < 2e-13 * np.abs(g["ovl_cross"]).max()
    io = run(p1, po, [(i, 0) for i in range(3)])
    assert np.abs(io - g["ovl_io"][:, 0]).max() < 2e-13 * np.abs(g["ovl_io"]).max()
    cc = run(po, po, [(0, 0)])
    assert abs(cc[0, ns // 2, ns // 2] - g["outovlc"][0]) < 1e-13 * g["outovlc"][0]


@pytest.mark.parametrize("name,nst", [("tiny", 3), ("small", 2), ("smallm", 2)])
def test_resident_path_vs_oracle(name, nst):
    from pyimcom_amd import synth
    from tests import parity as smoke

    rep = smoke.check_batch(synth.CONFIGS[name], n_stamps=nst, verbose=True)
    assert rep["tables"] < smoke.TOL["tables"]


def test_resident_ragged_exposures():
    """Variable exposure depth per stamp (cfg-4 style): ragged N inside one batch."""
    import dataclasses

    from pyimcom_amd import synth
    from tests import parity as smoke

    cfg = dataclasses.replace(synth.CONFIGS["small"], name="smallr", n_expo=(2, 5), fade=0)
    smoke.check_batch(cfg, n_stamps=4, first_id=40, verbose=True)


def test_smoke_entry():
    import __graft_entry__ as g

    g.smoke()


@pytest.mark.parametrize("kernel", ["Iterative", "Empirical"])
def test_resident_path_secondary_kernels(kernel):
    """The device-resident stamp path with the Iterative / Empirical LA kernels (lakernel.py:533-805) vs the oracle."""
    import dataclasses

    from pyimcom_amd import synth
    from tests import parity as smoke

    cfg = dataclasses.replace(synth.CONFIGS["tiny"], kernel=kernel)
    if kernel == "Iterative":
        # kappa/C = 0.1 (cond ~ 1e2): CG converges below rtol well inside iter_max.  At the tiny config's own 6e-4
        # (cond 1.7e4) it runs into the 30-step limit and the iterate it stops on amplifies 1e-16 differences in A
        # to per cent level -- nothing a parity test can pin (tests/test_gpu_iter_empir.py treats that regime).
        cfg = dataclasses.replace(cfg, kappaC=(0.1,))
    rep = smoke.check_batch(cfg, n_stamps=3)
    assert rep["stamp0"]["T"] < (2e-4 if kernel == "Iterative" else 1e-6)


def test_resident_path_many_input_layers():
    """n_inframe = 6 (science + noise / injected layers, coadd.py:975): the epilogue walks T once per four layers."""
    import dataclasses

    from pyimcom_amd import synth
    from tests import parity as smoke

    cfg = dataclasses.replace(synth.CONFIGS["small"], n_inframe=6)
    smoke.check_batch(cfg, n_stamps=2)


def test_resident_path_more_layers_than_exposures():
    """Two exposures, six input layers: the epilogue's LDS slots are sized by the LARGER of the two counts (the frames' sums of a pass
    -- four layers -- meet in the slots the per-exposure sums have left), with and without the fade taper."""
    import dataclasses

    from pyimcom_amd import synth
    from tests import parity as smoke

    for fade in (0, 2):
        cfg = dataclasses.replace(synth.CONFIGS["small"], name=f"small_l6e2f{fade}", n_inframe=6, n_expo=2, fade=fade)
        smoke.check_batch(cfg, n_stamps=2)


def test_two_target_psfs_vs_oracle():
    """n_out = 2 (OUTPSF plus one cfg.outpsf_extra entry): the reference solves every target on its own with
    kappa = kappaC * C of that target (lakernel.py:121-128); A is shared.  Every output of both targets against the
    oracle, and the two targets must really differ."""
    import dataclasses

    import torch

    from pyimcom_amd import synth
    from tests import parity as smoke
    from pyimcom_amd.stamps import PSFGroupTables, StampBatch

    cfg = dataclasses.replace(synth.CONFIGS["tiny"], name="tiny2", n_out=2, fade=1)
    rep = smoke.check_batch(cfg, n_stamps=2, verbose=True)
    assert "stamp1.target1" in rep
    stamps = [synth.make_stamp(cfg, i) for i in range(2)]
    psfs, target = synth.make_psfs(cfg, max(s.n_expo for s in stamps))
    tabs = PSFGroupTables(psfs, target, cfg.nfft)
    assert tabs.n_out == 2 and abs(tabs.Cs[0] - tabs.Cs[1]) > 0.05 * tabs.Cs[0]
    sb = StampBatch(cfg, stamps, tabs)
    sb.run()
    torch.cuda.synchronize()
    r0, r1 = sb.results()
    assert (r0.outimage - r1.outimage).abs().max() > 1e-3 * r0.outimage.abs().max()
    # target 0 of the pair == the single-target run, bit for bit
    one = StampBatch(cfg, stamps, PSFGroupTables(psfs, target[:1], cfg.nfft)).run()
    torch.cuda.synchronize()
    for name in ("UC", "Sigma", "kappa", "outimage", "Tsum_inpix", "Neff"):
        assert torch.equal(getattr(one, name), getattr(r0, name)), name


def test_empty_stamp_in_resident_batch():
    """A stamp without input pixels inside a resident batch (block corners with every neighbour masked): the
    reference's N == 0 case (lakernel.py:110-119: UC = 1, Sigma = 0, kappa = 1, T empty) next to a normal stamp,
    which must come out exactly as it does alone."""
    import torch

    from pyimcom_amd import synth
    from pyimcom_amd.stamps import PSFGroupTables, StampBatch

    cfg = synth.CONFIGS["tiny"]
    st = synth.make_stamp(cfg, 3)
    psfs, target = synth.make_psfs(cfg, st.n_expo)
    tabs = PSFGroupTables(psfs, target, cfg.nfft)
    alone = StampBatch(cfg, [st], tabs).run()
    torch.cuda.synchronize()
    ldn = alone.Tt.shape[1]
    dev = alone.Tt.device
    x = torch.zeros((2, ldn), dtype=torch.float64, device=dev)
    y = torch.zeros_like(x)
    expo = torch.zeros((2, ldn), dtype=torch.int32, device=dev)
    indata = torch.zeros((2, cfg.n_inframe, ldn), dtype=torch.float32, device=dev)
    x[1, : st.n], y[1, : st.n] = torch.as_tensor(st.x, device=dev), torch.as_tensor(st.y, device=dev)
    expo[1, : st.n] = torch.as_tensor(st.expo, device=dev)
    indata[1, :, : st.n] = torch.as_tensor(st.indata, device=dev)
    sb = StampBatch.from_device(cfg, tabs, [0, st.n], x, y, expo, indata, [st.out_x0] * 2, [st.out_y0] * 2, st.n_expo)
    res = sb.run()
    torch.cuda.synchronize()
    from oracle import oracle as orc

    ones = np.ones((cfg.n2f, cfg.n2f), np.float32)
    orc.trapezoid(ones, cfg.fade)  # the maps are tapered like any other stamp's (coadd.py:1118-1122)
    assert np.array_equal(res.UC[0].cpu().numpy(), ones) and np.array_equal(res.kappa[0].cpu().numpy(), ones)
    assert torch.all(res.Sigma[0] == 0)
    assert torch.all(res.outimage[0] == 0) and torch.all(res.Tsum_inpix[0] == 0) and torch.all(res.Tsum_stamp[0] == 0)
    assert torch.all(res.Tt[0] == 0)
    for name in ("UC", "Sigma", "kappa", "outimage", "Tsum_inpix", "Neff", "Tsum_stamp"):
        assert torch.equal(getattr(res, name)[1], getattr(alone, name)[0]), name


def test_alternative_code_paths_in_subprocess():
    """The switches kept for A/B runs and as cross-checks must not rot: the unfused diagonal-block launches
    (IMCOM_SOLVE_UNFUSED), the dense-DFT table path (IMCOM_PSF_OVERLAP=gemm) and -- in the developer build of the library
    (make DEV=1 -> libimcom_hip_dev.so, selected by IMCOM_HIP_LIB; built by __graft_entry__.build()) -- the Jacobi eigensolver
    (IMCOM_EIGH=jacobi) and the experimental LDS-window A builder (IMCOM_BUILD_A=window)
    are read once per process, so the parity check runs in a child process with all of them set."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IMCOM_SOLVE_UNFUSED="1", IMCOM_PSF_OVERLAP="gemm", PYTHONPATH=root)
    dev_lib = os.path.join(root, "pyimcom_amd", "lib", "libimcom_hip_dev.so")
    # __graft_entry__.build() makes the developer library: its absence is a broken build, not a reason to pass with half the test
    assert os.path.exists(dev_lib), f"{dev_lib} is missing: run `make -C pyimcom_amd/csrc DEV=1` (or __graft_entry__.build())"
    env.update(IMCOM_HIP_LIB=dev_lib, IMCOM_EIGH="jacobi", IMCOM_BUILD_A="window")
    # one Eigen batch with its reflector products beside the reduction on the second queue (the default below 24 stamps), that queue
    # confined to 192 CUs (IMCOM_AUX_CUS: read when the context is created); the sub-batch variants: tests/test_gpu_eigen_indef.py
    env.update(IMCOM_EIGEN_SPLIT="1", IMCOM_EIGEN_OVERLAP="1", IMCOM_AUX_CUS="192")
    env.update(IMCOM_LARFT="serial")   # the block reflectors' triangular factors column by column (default: MFMA triangular inverse)
    code = ("import dataclasses; from pyimcom_amd import synth; from tests import parity as smoke; "
            "smoke.check_batch(synth.CONFIGS['small'], 2); "
            "smoke.check_batch(dataclasses.replace(synth.CONFIGS['tiny'], kernel='Eigen'), 2); print('alt paths ok')")
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "alt paths ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    # IMCOM_FFT_COLS_SPLIT: the tables' column transform as independent workgroups of four waves (same arithmetic: the static sizes'
    # tables against numpy, and the resident path on them); IMCOM_EPI_LDS_PAD: the epilogue at one workgroup per CU
    env = dict(os.environ, IMCOM_FFT_COLS_SPLIT="1", IMCOM_EPI_LDS_PAD="45000", PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "tests/test_gpu_psfs.py", "tests/test_gpu_stamps.py", "-k",
                          "(mixed_radix and (32-8 or 24-16 or 32-16)) or test_resident_path_vs_oracle or windows"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and " passed" in out.stdout and "failed" not in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    # IMCOM_FFT_GENERIC: the general line-FFT kernels also at the sizes the static 16 x 16 x r kernels normally take
    env = dict(os.environ, IMCOM_FFT_GENERIC="1", PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "tests/test_gpu_psfs.py", "-k", "mixed_radix and (32-8 or 24-16)"],
                         env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "2 passed" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("name,kw,scale", [
    ("one_frame", dict(n_inframe=1), 1.0), ("one_expo", dict(n_expo=1), 1.0), ("odd_n2_nofade", dict(n2=7, fade=0), 1.0),
    ("fade3", dict(fade=3), 1.0), ("wide_pad", dict(inpad_as=0.3), 1.0), ("multi_kappa4", dict(kappaC=(1e-5, 1e-4, 1e-3, 1e-2)), 1.0),
    ("multi_kappa5", dict(kappaC=(1e-6, 1e-5, 1e-4, 1e-3, 1e-2)), 10.0), ("airy", dict(psf="airy"), 1.0),
    ("no_penalty", dict(flat_penalty=0.0), 1.0), ("eigen_multi", dict(kernel="Eigen", kappaC=(1e-5, 1e-2)), 1.0),
    ("empirical", dict(kernel="Empirical"), 1.0), ("two_targets_eigen", dict(kernel="Eigen", n_out=2), 1.0),
    ("many_expo", dict(n_expo=9), 1.0)])
def test_unusual_configurations(name, kw, scale):
    """Corners of the configuration space on the resident path against the oracle: one input layer, one exposure, odd
    stamp size without fade, a wide fade, a large acceptance radius, four and five kappa nodes (the latter with the wider
    bound on T that smoke.check_batch documents), Airy PSFs, no flat penalty, Eigen with two nodes / two targets, the
    empirical kernel, nine exposures."""
    import dataclasses

    from pyimcom_amd import synth
    from tests import parity as smoke

    smoke.check_batch(dataclasses.replace(synth.CONFIGS["tiny"], name=name, **kw), n_stamps=3, tolT_scale=scale)


@pytest.mark.parametrize("kw", [dict(), dict(kappaC=(1e-5, 1e-4, 1e-3)), dict(n_out=2), dict(n_expo=9, n_inframe=3)])
def test_coaddition_inside_the_solve_equals_the_stand_alone_epilogue(kw, monkeypatch):
    """With fade 0 the Cholesky solve and the coaddition can be one call (imcom_solve_chol_resident_coadd, IMCOM_EPILOGUE_FUSED=1): with one kappa node the
    per-exposure sums and T . indata are taken from the tiles of T inside the backward launches.  The stand-alone epilogue
    (imcom_coadd_epilogue), run afterwards on the T the solve left, must give the same coaddition -- float64 sums of the same
    float32 T in another order: images to a float32 ulp, weight sums to 1e-12 -- for a ragged batch, several
    kappa nodes (the epilogue then runs inside the call), two target PSFs, nine exposures / three input frames."""
    import dataclasses

    import torch

    from pyimcom_amd import synth
    from pyimcom_amd.stamps import PSFGroupTables, StampBatch

    monkeypatch.setenv("IMCOM_EPILOGUE_FUSED", "1")
    cfg = dataclasses.replace(synth.CONFIGS["smallm"], name="fusedco", **dict(dict(kappaC=(6e-4,)), **kw))
    assert cfg.fade == 0
    stamps = [synth.make_stamp(cfg, 300 + i) for i in range(5)]
    psfs, target = synth.make_psfs(cfg, max(s.n_expo for s in stamps))
    sb = StampBatch(cfg, stamps, PSFGroupTables(psfs, target, cfg.nfft))
    assert sb.n_out == cfg.n_out
    sb.build()
    sb.solve()
    assert sb._coadded == set(range(cfg.n_out))  # the call coadded every target
    sb.coadd()
    torch.cuda.synchronize()
    names = ("outimage_o", "Tsum_stamp_o", "Tsum_inpix_o", "Neff_o")
    fused = {k: getattr(sb, k).clone() for k in names}
    assert sb._coadded == set()  # (coadd() found the targets done and reset the marks)
    sb.coadd()  # nothing is marked now: the stand-alone epilogue on the same T
    torch.cuda.synchronize()
    for k in names:
        a, b = fused[k].double(), getattr(sb, k).double()
        tol = 2e-7 if k == "outimage_o" else 1e-12
        assert float((a - b).abs().max()) <= tol * float(b.abs().max()), k
    assert float(sb.outimage_o.abs().max()) > 0


def test_solve_in_two_halves_equals_the_synchronous_solve():
    """imcom_solve_chol_resident_begin / _end (StampBatch.solve_begin / solve_end: what blockrun.coadd_block runs, with the next pass
    prepared in between): the same bits as the synchronous entry -- for a normal batch, with the fade taper, and for a batch in which
    one stamp's A + kappa I is not positive definite: _end then reports the failure, the synchronous entry repairs it as the reference
    does (lakernel.py:262-279) and info names the stamp."""
    import dataclasses

    import torch

    from pyimcom_amd import synth
    from pyimcom_amd.stamps import PSFGroupTables, StampBatch

    for fade, breakit in ((0, False), (2, False), (0, True)):
        cfg = dataclasses.replace(synth.CONFIGS["small"], name=f"small_halves_{fade}_{int(breakit)}", fade=fade)
        stamps = [synth.make_stamp(cfg, i) for i in range(3)]
        psfs, target = synth.make_psfs(cfg, max(s.n_expo for s in stamps))
        tabs = PSFGroupTables(psfs, target, cfg.nfft)
        out = []
        for halves in (False, True):
            sb = StampBatch(cfg, stamps, tabs)
            sb.build()
            if breakit:  # stamp 1: a diagonal entry far below zero -- the factorisation fails, the eigh-shift repair takes over
                sb.A[1, 5, 5] -= 10.0 * float(sb.A[1].diagonal().abs().max())
            if halves:
                sb.solve_begin()
                torch.zeros(1 << 20, device="cuda:0").sum()  # (other work queued between the halves)
                sb.solve_end()
            else:
                sb.solve()
            sb.coadd()
            torch.cuda.synchronize()
            r = sb.result()
            out.append({k: getattr(r, k).clone() for k in ("Tt", "UC", "Sigma", "kappa", "outimage", "Neff")} | {"info": sb.info_o[0].copy()})
        a, b = out
        assert (a["info"] == b["info"]).all() and (int(a["info"][1]) != 0) == breakit and a["info"][0] == 0
        for k in ("Tt", "UC", "Sigma", "kappa", "outimage", "Neff"):
            assert torch.equal(a[k], b[k]), (fade, breakit, k)


def test_solve_retries_when_the_workspace_cannot_grow(monkeypatch):
    """The library's workspace is a device allocation of its own: memory torch's caching allocator holds unused is not available to
    it.  IMCOM_ERR_NOMEM from the solve makes StampBatch hand that memory back and try once more; another status is raised."""
    import torch

    from pyimcom_amd import synth
    from pyimcom_amd._lib import ImcomError
    from pyimcom_amd.stamps import PSFGroupTables, StampBatch

    cfg = synth.CONFIGS["tiny"]
    stamps = [synth.make_stamp(cfg, i) for i in range(2)]
    psfs, target = synth.make_psfs(cfg, max(s.n_expo for s in stamps))
    sb = StampBatch(cfg, stamps, PSFGroupTables(psfs, target, cfg.nfft))
    ref = sb.run()
    torch.cuda.synchronize()
    T_ref = ref.Tt.clone()
    real, calls = StampBatch._solve_target, []

    def flaky(self, *a, **k):
        calls.append(1)
        if len(calls) == 1:
            raise ImcomError(-3, "device workspace: out of memory (injected)")
        return real(self, *a, **k)

    monkeypatch.setattr(StampBatch, "_solve_target", flaky)
    sb.build()
    sb.solve()
    sb.coadd()
    torch.cuda.synchronize()
    assert len(calls) == 2 and torch.equal(sb.result().Tt, T_ref)
    calls.clear()
    monkeypatch.setattr(StampBatch, "_solve_target", lambda self, *a, **k: (_ for _ in ()).throw(ImcomError(-5, "injected")))
    with pytest.raises(ImcomError):
        sb.solve()


@pytest.mark.parametrize("block_of", [128, 16, -16])
@pytest.mark.parametrize("shifts_in_kappa", [(3.0, 0.0, 40.0, 1.5), (0.0, 0.0, 0.0, 2.0, 0.0)])
def test_repair_of_large_stamps_by_the_subspace_iteration_vs_oracle(shifts_in_kappa, block_of, monkeypatch):
    """The Cholesky repair (lakernel.py:262-279: AA_ii += |w[0]| + 1e-16, w[0] the smallest eigenvalue of A) at a size where the library
    finds w[0] WITHOUT an eigendecomposition (api.hip lambda_min_subspace: trial factorisations, subspace iteration with the inverse
    on 16 vectors (lmin_skinny.hip; the sweeps as two launches per block row, the form few stamps take, or -- block_of = -16 -- as one
    workgroup per stamp, the form of a pass of more than 128 stamps) or on 128 (the form of rounds 5 / 6a, IMCOM_LMIN_SKINNY=0), Rayleigh-Ritz
    with A; matrices of 1024 rows and more).  cfg-2 stamps (N ~ 2.2k), a batch of four of which three
    are made indefinite by different amounts -- A - c I with c a multiple of kappa, so that w[0] = lambda_min(A) - c sits in the dense
    lower end of a real PSF-overlap spectrum -- and one stays as it is.  T, the maps and info against the oracle's CholKernel (numpy eigh
    + scipy cholesky), through the synchronous entry and through begin / end / redo (only the failed stamps are solved again: the healthy
    stamp's outputs must come out bit for bit as a batch without failures gives them)."""
    import torch

    from oracle import oracle as orc
    from pyimcom_amd import synth
    from pyimcom_amd.stamps import PSFGroupTables, StampBatch

    monkeypatch.setenv("IMCOM_LMIN_SKINNY", "0" if block_of == 128 else "1")
    if block_of == -16:
        monkeypatch.setenv("IMCOM_LMIN_FEW_MAX", "0")
    cfg = synth.CONFIGS["cfg2"]
    stamps = [synth.make_stamp(cfg, 40 + i) for i in range(len(shifts_in_kappa))]
    psfs, target = synth.make_psfs(cfg, cfg.n_expo)
    tabs = PSFGroupTables(psfs, target, cfg.nfft)
    kap = cfg.kappaC[0] * tabs.C
    shifts = [f * kap for f in shifts_in_kappa]  # (the second case: ONE failure in a batch -- the iteration's products then run per wanted stamp)
    failed = [f != 0.0 for f in shifts_in_kappa]
    ref = StampBatch(cfg, stamps, tabs)
    ref.run()
    torch.cuda.synchronize()
    healthy = {k: getattr(ref, k)[1].clone() for k in ("Tt", "UC", "Sigma", "kappa", "outimage")}
    assert not ref.info.any()
    want = None
    w0_abs = None
    for mode in ("synchronous", "halves", "expected", "hint", "hint_small", "hint_large", "hint_far"):
        sb = StampBatch(cfg, stamps, tabs)
        sb.build()
        for s, c in enumerate(shifts):
            if c:
                n = int(sb.n[s])
                sb.A[s, :n, :n].diagonal().sub_(c)
        if want is None:
            want = []
            for s, st in enumerate(stamps):
                n, m = st.n, cfg.m
                A = sb.A[s, :n, :n].cpu().numpy()
                mB = np.ascontiguousarray(sb.Bt[s, :n, :m].cpu().numpy().T)
                want.append(orc.chol_kernel(A, mB, tabs.C, np.array(cfg.kappaC), cfg.uctarget, cfg.sigmamax) + (np.linalg.eigvalsh(A),))
        if mode == "synchronous":
            sb.solve()
            sb.coadd()
        elif mode.startswith("hint"):
            # the driver's estimate of max |w[0]| (what the pass before found: StampBatch.repair_absmax): the iteration starts there.  A good one,
            # one that is too small for every failed stamp (their first factorisation fails: back to the shift they would have started with)
            # one that is three times too large (a slow first shift: a second factorisation, as without a hint) and one from another regime
            # (twenty times: the shift is replaced after the first Rayleigh-Ritz step by 2 (|theta| + |r|)) -- the same answers
            f = {"hint": 1.0, "hint_small": 0.5, "hint_large": 3.0, "hint_far": 20.0}[mode]
            sb.solve_begin(expect_repair=True, repair_hint=f * w0_abs)
            assert sb.solve_end() is False and abs(sb.repair_share - np.mean(failed)) < 1e-12
            assert sum(failed) <= sb.ctx.last_repair()[0] <= len(stamps)  # (the repaired stamps' w[0]; a healthy one is only DECIDED to be positive definite)
            sb.coadd()
        elif mode == "expected":
            # a driver that has seen the previous pass repaired throughout skips the factorisation that fails: every stamp is handed over
            # as "known to fail", and the smallest eigenvalue itself says which stamps are positive definite after all (the healthy ones
            # here): those are factored plainly, info = 0, as the reference's cholesky() would have succeeded
            sb.solve_begin(expect_repair=True)
            assert sb.solve_end() is False and abs(sb.repair_share - np.mean(failed)) < 1e-12
            sb.coadd()
        else:
            sb.solve_begin()
            sb.coadd()
            again = sb.solve_end()
            assert again is not False and list(again) == failed
            sb.coadd(only=again)
        torch.cuda.synchronize()
        if mode == "expected":
            cnt, lo, hi = sb.ctx.last_repair()
            # (all stamps went into the iteration -- the healthy ones until lambda_min >= theta - |r| had decided that A + kappa I is positive
            # definite; the most negative of the smallest eigenvalues is LAPACK's)
            assert sum(failed) <= cnt <= len(stamps) and abs(lo - min(w[5][0] for w in want)) <= 1e-9 * abs(lo)
            w0_abs = sb.repair_absmax
            assert w0_abs == max(abs(lo), abs(hi))
        r = sb.result()
        assert list(r.info) == [int(f) for f in failed], (mode, r.info)
        for s, st in enumerate(stamps):
            To, Uo, So, ko, info_o, lam = want[s]
            assert int(r.info[s]) == info_o
            # after the repair the smallest eigenvalue of the factored matrix is kappa + 1e-16: cond = (lam_max + kappa + |w0|) / kappa
            cond = (lam[-1] + kap + abs(min(lam[0], 0.0))) / kap
            T = r.T(s).cpu().numpy()
            assert np.abs(T - To).max() <= (1e-6 + 50 * cond * 2.2e-16) * np.abs(To).max(), (mode, s)
            s2 = (cfg.n2f, cfg.n2f)
            assert np.allclose(r.UC[s].cpu().numpy(), Uo.reshape(s2), rtol=1e-5 + 50 * cond * 2.2e-16, atol=1e-9), (mode, s)
            assert np.allclose(r.Sigma[s].cpu().numpy(), So.reshape(s2), rtol=1e-5 + 50 * cond * 2.2e-16, atol=1e-9), (mode, s)
        for k, v in healthy.items():  # the stamp without a failure: what a batch without failures gives, bit for bit
            assert torch.equal(getattr(sb, k)[1], v), (mode, k)  # (also in the "expected" mode: its plain factorisation is the same launches on the same data)
