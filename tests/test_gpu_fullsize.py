"""GPU parity at BASELINE.json's full sizes: every config's stamps run through the device-resident path and
stamp 0..k of each batch is checked against the CPU oracle (A, B, T, maps, coaddition), plus size-independent
properties on the whole batch.  Run with -m gpu on an MI355X (the oracle needs a few seconds per stamp)."""

import dataclasses

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,nst", [("cfg1", 1), ("cfg2", 2), ("cfg2f", 1), ("cfg4", 3)])
def test_baseline_config_vs_oracle(name, nst):
    from pyimcom_amd import smoke, synth

    smoke.check_batch(synth.CONFIGS[name], n_stamps=nst, verbose=True)


def test_cfg3_eigen_sweep_vs_oracle():
    """cfg-3: eigendecomposition kappa sweep at 8 exposures; reduced to a 24x24-output stamp so that the Jacobi
    eigensolver and the oracle's eigh finish in seconds (same code path as N ~ 2.9k)."""
    from pyimcom_amd import smoke, synth

    cfg = dataclasses.replace(synth.CONFIGS["cfg3"], name="cfg3s", n2=24, inpad_as=0.3)
    smoke.check_batch(cfg, n_stamps=2, verbose=True)


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
