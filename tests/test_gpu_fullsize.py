"""GPU parity at BASELINE.json's full sizes: every config's stamps run through the device-resident path and
stamp 0..k of each batch is checked against the CPU oracle (A, B, T, maps, coaddition), plus size-independent
properties on the whole batch.  Run with -m gpu on an MI355X (the oracle needs a few seconds per stamp)."""

import dataclasses

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,nst", [("cfg1", 1), ("cfg2", 2), ("cfg2f", 1), ("cfg4", 3), ("cfg5", 1), ("paper4", 1)])
def test_baseline_config_vs_oracle(name, nst):
    """Every Cholesky configuration of BASELINE.json at its full stamp size -- cfg-5 included: ONE deep-field stamp, 16 exposures,
    N ~ 5.9k (lakernel.py:281-323 on a 5.9k system: the oracle needs about half a minute on the box's host cores) -- A, B, T,
    maps and image against the oracle, from the device's tables and (stamp 0) from the oracle's own."""
    from pyimcom_amd import synth
    from tests import parity as smoke

    smoke.check_batch(synth.CONFIGS[name], n_stamps=nst, verbose=True)


def test_cfg3_eigen_sweep_small_vs_oracle():
    """cfg-3 reduced to a 24x24-output stamp (N ~ 0.9k): the quick variant of the full-size test below."""
    from pyimcom_amd import synth
    from tests import parity as smoke

    cfg = dataclasses.replace(synth.CONFIGS["cfg3"], name="cfg3s", n2=24, inpad_as=0.3)
    smoke.check_batch(cfg, n_stamps=2, verbose=True)


def test_cfg3_full_vs_oracle():
    """cfg-3 at BASELINE size: 8 exposures, 48x48 outputs, N ~ 2.9k, Eigen kernel with the three-node kappa sweep
    (lakernel.py:174-223), a batch of two stamps, every output against the oracle (numpy eigh + the C lakernel1).
    Exercises the size-dependent branches of the tridiagonal QR solver (multishift chase, pipelined bulges, deflation,
    the rotation-log ring).  Also the decision accounting SURVEY 8d asks for: a flipped bisection decision of
    lakernel1 moves kappa by at least (kCmax/kCmin)^(2^-14) - 1 = 2.8e-4, so any pixel whose kappa differs from the
    oracle's by more than 1e-6 relative is a flip; they are counted and must be (near-)ties of the U/C target."""
    from pyimcom_amd import synth
    from tests import parity as smoke

    cfg = synth.CONFIGS["cfg3"]
    rep = smoke.check_batch(cfg, n_stamps=2, verbose=True, collect=("kappa", "UC"))
    flips = 0
    for key, r in rep.items():
        if not key.startswith("stamp"):
            continue
        assert 2700 < r["n"] < 3100, r["n"]
        k_gpu, k_ref = r["kappa_gpu"].astype(np.float64), r["kappa_ref"].astype(np.float64)
        flipped = np.abs(k_gpu / k_ref - 1.0) > 1e-6
        flips += int(flipped.sum())
        # a legitimate flip is a tie: the pixel's U/C sits on the target within rounding of the eigen-sums
        assert np.all(np.abs(r["UC_ref"][flipped] - cfg.uctarget) <= 1e-9), "kappa decision flipped away from a tie"
    print(f"[cfg3] lakernel1 bisection decisions flipped vs the oracle: {flips} of {2 * cfg.m} pixels x 13 steps")
    assert flips <= 2


@pytest.mark.parametrize("n", [1000, 2944])
def test_eigh_on_cfg3_matrix(n):
    """imcom_eigh at the sizes the cfg-3 path meets, on a real PSF-overlap matrix (the A of a cfg-3 stamp built on the
    device; the leading n x n block for n = 1000, the identity-padded 2944 block otherwise) against LAPACK:
    |dlam| <= 2e-14 |A|_2, residual and orthogonality at the 1e-13 level."""
    import ctypes as C

    import torch

    from pyimcom_amd import synth
    from pyimcom_amd._lib import MEM_DEVICE, check, lib
    from pyimcom_amd.stamps import PSFGroupTables, StampBatch

    cfg = synth.CONFIGS["cfg3"]
    st = synth.make_stamp(cfg, 3)
    psfs, target = synth.make_psfs(cfg, st.n_expo)
    sb = StampBatch(cfg, [st], PSFGroupTables(psfs, target, cfg.nfft), ldn=2944 if st.n <= 2944 else None)
    sb.build()
    torch.cuda.synchronize()
    n_act = min(n, st.n)
    A = sb.A[0, :n, :n].contiguous() if n <= st.n else sb.A[0, :n, :n].contiguous()
    if n > st.n:  # rows/columns beyond the stamp's pixels: identity (imcom_build_A's padding contract)
        pad = A[st.n:, :].cpu().numpy()
        assert np.array_equal(pad, np.eye(n)[st.n:]), "identity padding out to ldn"
    lam = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    Q = torch.zeros((n, n), dtype=torch.float64, device="cuda:0")
    ns = np.array([n], dtype=np.int32)
    sb._stream()
    check(lib.imcom_eigh(sb.ctx.handle, 1, ns.ctypes.data_as(C.c_void_p), n, C.c_void_p(A.data_ptr()), C.c_void_p(lam.data_ptr()),
                         C.c_void_p(Q.data_ptr()), MEM_DEVICE))
    torch.cuda.synchronize()
    Ah, lh, Qh = A.cpu().numpy(), lam.cpu().numpy(), Q.cpu().numpy()
    w = np.linalg.eigvalsh(Ah)
    norm = max(abs(w[0]), abs(w[-1]))
    e_lam = np.abs(lh - w).max() / norm
    e_res = np.abs(Ah @ Qh - Qh * lh).max() / norm
    e_orth = np.abs(Qh.T @ Qh - np.eye(n)).max()
    print(f"[eigh n={n} (stamp pixels {n_act})] |A|={norm:.3g} dlam/|A|={e_lam:.2e} resid/|A|={e_res:.2e} orth={e_orth:.2e}")
    assert np.all(np.diff(lh) >= 0)
    assert e_lam <= 2e-14 and e_res <= 2e-13 and e_orth <= 2e-13


def test_cfg5_properties():
    """cfg-5 (16 exposures, N ~ 5.9k): too big for a full oracle pass in seconds, so check the solve through
    properties: exact symmetry of A, (A + kappa I) T^T = B^T residual in float64 on the device, maps consistent
    with T, coaddition linear in the input frames, identical results when run twice."""
    import torch

    from pyimcom_amd import synth
    from pyimcom_amd.stamps import PSFGroupTables, StampBatch

    cfg = synth.CONFIGS["cfg5"]
    stamps = [synth.make_stamp(cfg, 500 + i) for i in range(2)]
    psfs, target = synth.make_psfs(cfg, 16)
    tabs = PSFGroupTables(psfs, target, cfg.nfft)
    b = StampBatch(cfg, stamps, tabs)
    r = b.run()
    torch.cuda.synchronize()
    assert np.all(r.info == 0)
    for s, st in enumerate(stamps):
        n, m = st.n, cfg.m
        assert 5500 < n < 6300
        A = b.A[s, :n, :n]
        assert torch.equal(A, A.T)
        kap = cfg.kappaC[0] * tabs.C
        T = r.Tt[s, :n, :m].double()                      # [n, m] = T^T
        Bt = b.Bt[s, :n, :m]
        res = A @ T + kap * T - Bt                        # torch GEMM as the independent checker
        # float32 storage of T limits the residual: |A| |dT| with |dT| <= 6e-8 |T|
        bound = 6e-8 * (A.abs() @ T.abs() + kap * T.abs()) + 1e-12
        assert bool((res.abs() <= 4 * bound).all())
        Sig = (T * T).sum(0)
        assert torch.allclose(r.Sigma[s].ravel().double(), Sig, rtol=1e-5, atol=1e-9)
        D = (Bt * T).sum(0)
        UC = 1.0 - (kap * Sig + D) / tabs.C
        assert torch.allclose(r.UC[s].ravel().double(), UC, rtol=0, atol=5e-6)
        assert float(r.kappa[s].min()) == float(r.kappa[s].max()) == np.float32(kap)
        img = torch.as_tensor(st.indata, device=T.device).double() @ T   # [n_inframe, m]
        scale = torch.as_tensor(np.abs(st.indata), device=T.device).double() @ T.abs()
        assert bool(((r.outimage[s].reshape(cfg.n_inframe, m).double() - img).abs() <= 2e-5 * scale + 1e-12).all())
    first = [x.clone() for x in (r.Tt, r.UC, r.outimage, r.Neff)]
    r2 = b.run()
    torch.cuda.synchronize()
    for x, y in zip(first, (r2.Tt, r2.UC, r2.outimage, r2.Neff)):
        assert torch.equal(x, y)  # deterministic: no atomics on the data path
    # linearity of the coaddition in the input frames
    b.indata.mul_(2.0)
    b.coadd()
    torch.cuda.synchronize()
    assert torch.allclose(b.outimage, 2.0 * first[2].reshape(b.outimage.shape), rtol=1e-6, atol=0)


def test_cfg2_block_groups_equal_single_group():
    """cfg-2 geometry through the block driver (selection, tables, A, B, Cholesky, coaddition, block maps) at full stamp
    size (N ~ 2.2k), as a size-independent property: when every 2x2 PSF group holds the SAME PSFs, the per-group route
    (BlockTables: self / cross / input-output sets, per-stamp pair maps, flipped and swapped tables) must reproduce the
    one-group route.  The tables of a pair come from different FFT evaluations on the two routes (1e-16 apart), so the
    maps agree to the conditioning of the solve, not bit for bit."""
    import torch

    from pyimcom_amd import synth
    from pyimcom_amd.blockrun import coadd_block
    from pyimcom_amd.select import InStampPool
    from pyimcom_amd.stamps import BlockTables, PSFGroupTables

    cfg = synth.CONFIGS["cfg2"]
    n1P, E = 3, cfg.n_expo
    inst = synth.make_instamps(cfg, n1P, E, np.random.default_rng(12))
    pool = InStampPool(inst, cfg.n_inframe)
    psfs, target = synth.make_psfs(cfg, E)
    one = coadd_block(cfg, pool, PSFGroupTables(psfs, target, cfg.nfft), n1P, E, batch=9)
    ng = (n1P + 3) // 2
    grp = coadd_block(cfg, pool, BlockTables({(gj, gi): psfs for gj in range(ng) for gi in range(ng)}, target, cfg.nfft, capacity=1500),
                      n1P, E, batch=5)
    torch.cuda.synchronize()
    a, b = one.out_map.cpu().numpy(), grp.out_map.cpu().numpy()
    assert np.isfinite(a).all() and np.abs(a).max() > 0
    assert np.abs(a - b).max() <= 1e-5 * np.abs(a).max()
    for k in ("UC", "Sigma", "kappa", "Tsum", "Neff"):
        x, y = one.maps[k].cpu().numpy(), grp.maps[k].cpu().numpy()
        assert np.allclose(x, y, rtol=1e-4, atol=1e-7 * np.abs(x).max()), k
    assert np.allclose(one.T_weightmap.cpu().numpy(), grp.T_weightmap.cpu().numpy(), rtol=1e-5, atol=1e-8)
    # cells=True: only the half (quarter) of a cross table that separations between two grid cells of InStamps can reach is
    # computed.  The arena is filled with NaN first: a single interpolation outside the computed part would show in the maps --
    # which must come out bit for bit as with whole tables.
    tabs = BlockTables({(gj, gi): psfs for gj in range(ng) for gi in range(ng)}, target, cfg.nfft, capacity=1500, cells=True)
    tabs.tables[1:].fill_(float("nan"))
    cel = coadd_block(cfg, pool, tabs, n1P, E, batch=5)
    torch.cuda.synchronize()
    part = tabs.tables[1 : tabs.used + 1]
    assert bool(torch.isnan(part).any()), "nothing was pruned"
    assert torch.equal(cel.out_map, grp.out_map)
    for k in ("UC", "Sigma", "kappa", "Tsum", "Neff"):
        assert torch.equal(cel.maps[k], grp.maps[k]), k


@pytest.mark.parametrize("kC", [(6e-4,), (1e-5, 1e-4, 1e-3)])
def test_kernel_class_seam_full_size_vs_oracle(kC):
    """ONE cfg-2 stamp (N ~ 2.2k, m = 2304) through the drop-in kernel class over host buffers -- the small-batch path of the
    library: split-K partial products in the Cholesky updates and the solves (8 parts per tile), the kappa nodes of a multi-kappa call as one
    node-major pass -- against the oracle's CholKernel on the same A, -B/2, C."""
    import torch

    from oracle import oracle as orc
    from pyimcom_amd import synth
    from pyimcom_amd.lakernel import HipCholKernel
    from pyimcom_amd.stamps import PSFGroupTables, StampBatch
    from tests.golden.make_golden import make_outst

    cfg = synth.CONFIGS["cfg2"]
    st = synth.make_stamp(cfg, 11)
    psfs, target = synth.make_psfs(cfg, st.n_expo)
    tabs = PSFGroupTables(psfs, target, cfg.nfft)
    sb = StampBatch(cfg, [st], tabs)
    sb.build()
    torch.cuda.synchronize()
    n, m = st.n, cfg.m
    A = sb.A[0, :n, :n].cpu().numpy().copy()
    mB = np.ascontiguousarray(sb.Bt[0, :n, :m].cpu().numpy().T)[None]
    C = np.array([tabs.C])
    kCa = np.array(kC)
    o = make_outst(A.copy(), mB.copy(), C, cfg.n2f, kCa, cfg.uctarget, cfg.sigmamax)
    K = HipCholKernel(o)
    K()
    assert np.array_equal(o.sysmata, A) and int(K.info[0]) == 0
    To, Uo, So, ko, _ = orc.chol_kernel(A, mB[0], float(C[0]), kCa, cfg.uctarget, cfg.sigmamax)
    lam = np.linalg.eigvalsh(A)
    cond = (lam[-1] + kCa[0] * C[0]) / (max(lam[0], 0.0) + kCa[0] * C[0])
    scale = 1.0 if len(kC) == 1 else 4.0  # the nv x nv reduced systems of build_reduced_T amplify the solve's rounding (parity.check_batch)
    assert np.abs(o.T[0] - To).max() <= (1e-6 + 50 * cond * 2.2e-16) * scale * np.abs(To).max()
    s2 = (cfg.n2f, cfg.n2f)
    assert np.allclose(o.UC[0], Uo.reshape(s2), rtol=1e-5 + 50 * cond * 2.2e-16, atol=1e-9)
    assert np.allclose(o.Sigma[0], So.reshape(s2), rtol=1e-5 + 50 * cond * 2.2e-16, atol=1e-9)
    assert np.allclose(o.kappa[0], ko.reshape(s2), rtol=1e-5, atol=0)
