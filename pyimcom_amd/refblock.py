"""Block seam: ``Block.coadd_output_stamps`` of the reference on the GPU, taking the reference's own containers.

    from pyimcom_amd.refblock import coadd_output_stamps
    coadd_output_stamps(blk, PSFGrp)      # instead of the stamp loop of coadd.py:2003-2084

``blk`` is a (duck-typed) ``pyimcom.coadd.Block`` after its constructor has parsed the configuration, loaded the input
images and partitioned their pixels (coadd.py:1560-1937); what is read from it:

    blk.cfg            n1P, n2, fade_kernel, n_inframe, dtheta [deg], instamp_pad [rad], linear_algebra, kappaC_arr,
                       uctarget, sigmamax, psf_circ, psf_norm, amp_penalty, outpsf, sigmatarget, use_filter,
                       outpsf_extra / sigmatarget_extra, postage_pad, iter_rtol / iter_max (Iterative), psfsplit, psf_interp
    blk.instamps[j][i] x_val, y_val, data, pix_count, pix_cumsum; psf_compute_point_pix on the even-even ones (coadd.py:682-714)
    blk.inimages[e]    get_psf_pos(world, use_shortrange=True), outpix2world2inpix(xy)   (coadd.py:531-640)
    blk.outwcs         all_pix2world(xy, 0);   blk.n_inimage;   blk.pad_sides
    PSFGrp             the reference's class after PSFGrp.setup(): nsamp, oversamp, dscale, nfft, npixpsf (psfutil.py:568-613);
    flat_penalty       PSFOvl.flat_penalty (psfutil.py:1065-1089)

What is written is what the reference's loop leaves behind (coadd.py:1939-2001, 2163-2181): ``blk.out_map`` f32
[n_out, n_inframe, Ns, Ns], ``blk.UC_map / Sigma_map / kappa_map / Tsum_map / Neff_map`` f32 [n_out, Ns, Ns] and
``blk.T_weightmap`` f32 [n_out, n_inimage, n1P, n1P], boundary recovery included.  The host keeps what only it can do -- the
PSF file broker and the WCS evaluation of the PSF sampling positions -- everything else (PSF sampling, overlap tables per
2x2 group of InStamps, pixel selection, A, B, the LA kernel, coaddition, block accumulation) runs in libimcom_hip.

Configurations this build cannot serve raise ``ImcomError`` with status IMCOM_ERR_UNSUPPORTED instead of silently taking
another path: PSFINTERP "G4460" (the 8x8 interpolator of furry_parakeet: no source and no numerical test in the reference,
SURVEY 8c) and PSF splitting (``psfsplit``, psfutil.py:594-606 -- a different table pipeline).
"""

import numpy as np

from ._lib import ImcomError
from .synth import WorkloadConfig  # the plain configuration record of the stamp seam (nothing synthetic about it)

IMCOM_ERR_UNSUPPORTED = -4
ARCSEC = np.pi / 180.0 / 3600.0  # pyimcom.config.Settings.arcsec (config.py:85-98)


def check_supported(cfg):
    """Raise IMCOM_ERR_UNSUPPORTED for the reference configurations that have no device path."""
    interp = str(getattr(cfg, "psf_interp", "D5512") or "D5512").upper()
    if interp != "D5512":
        raise ImcomError(IMCOM_ERR_UNSUPPORTED, f"PSFINTERP = {interp!r}: only the D5512 (10x10) interpolator is built "
                         "(iG4460C has no in-tree source or numerical test to pin it against, coadd.py:1599-1601)")
    if getattr(cfg, "psfsplit", None):
        raise ImcomError(IMCOM_ERR_UNSUPPORTED, "PSFSPLIT is set: the split-PSF table pipeline (psfutil.py:594-606, 1178-1242) is not built")
    kernel = getattr(cfg, "linear_algebra", "Cholesky")
    if kernel not in ("Cholesky", "Eigen", "Iterative", "Empirical"):
        raise ImcomError(IMCOM_ERR_UNSUPPORTED, f"LAKERNEL = {kernel!r}")
    if int(getattr(cfg, "n1P", 0)) % 2:  # the reference asserts n1 % 2 == 0 (config.py:503): the PSF groups are pairs of InStamp rows / columns
        raise ImcomError(IMCOM_ERR_UNSUPPORTED, f"n1P = {cfg.n1P} is odd: PSF groups are 2 x 2 InStamps (psfutil.py:1803-1824)")


def stamp_config(cfg, psfgrp, n_inimage, flat_penalty, name="block"):
    """The stamp-seam configuration record of a reference Config."""
    targets = 1 + len(getattr(cfg, "outpsf_extra", []) or [])
    return WorkloadConfig(name, int(cfg.n2), int(cfg.fade_kernel), float(cfg.dtheta) * 3600.0, int(n_inimage),
                          float(cfg.instamp_pad) / ARCSEC, str(cfg.linear_algebra), tuple(float(k) for k in np.atleast_1d(cfg.kappaC_arr)),
                          npixpsf=int(psfgrp.npixpsf), oversamp=int(psfgrp.oversamp), uctarget=float(cfg.uctarget),
                          sigmamax=float(cfg.sigmamax), flat_penalty=float(flat_penalty), n_inframe=int(cfg.n_inframe), n_out=targets,
                          no_qlt_ctrl=str(cfg.linear_algebra) == "Empirical" and bool(getattr(cfg, "no_qlt_ctrl", False)))  # coadd.py:856-858


class _HostAhead:
    """The host half of a PSF group -- the reference's own work: the PSF file broker (InImage.get_psf_pos) and the WCS evaluation
    of the nsamp^2 sampling positions (outpix2world2inpix, psfutil.py:751-771) -- for the groups the coming batches will need,
    on worker threads while the GPU is busy with the current batch.  ``threads=1`` (default): ONE worker, which then makes every
    call into the block's objects (the main thread hands on-demand requests to it too), so objects that are not thread-safe
    -- wcslib structures behind astropy WCS -- are never entered from two threads; more workers only for callers whose
    ``inimages`` are (``bench.py``'s duck-typed block: plain numpy).  At most ``ahead`` groups are held ready (21 MB each at
    six exposures)."""

    def __init__(self, work, threads=1, ahead=48):
        from concurrent.futures import ThreadPoolExecutor

        self.work, self.ahead = work, max(1, int(ahead))
        self.pool = ThreadPoolExecutor(max_workers=max(1, int(threads)), thread_name_prefix="imcom-psf")
        self.futures, self.queue = {}, []

    def schedule(self, keys):
        """The order in which groups will first be needed (from the block's plan)."""
        self.queue = [k for k in dict.fromkeys(keys) if k not in self.futures]
        self._top_up()

    def _top_up(self):
        while self.queue and len(self.futures) < self.ahead:
            k = self.queue.pop(0)
            if k not in self.futures:
                self.futures[k] = self.pool.submit(self.work, k)

    def get(self, key):
        f = self.futures.pop(key, None)
        if f is None:  # not foreseen (a group that comes back after its spectra were dropped): still on a worker thread
            if key in self.queue:
                self.queue.remove(key)
            f = self.pool.submit(self.work, key)
        out = f.result()
        self._top_up()
        return out

    def close(self):
        for f in self.futures.values():
            f.cancel()
        self.pool.shutdown(wait=True)
        self.futures, self.queue = {}, []


LATTICE = 17  # nodes per axis of the lattice the sampling positions are evaluated on (positions="lattice")


def input_psf_groups(blk, psfgrp, device, ctx=None, host_threads=1, positions="lattice"):
    """PSFGrp._build_inpsfgrp for the 2x2 groups of InStamps (psfutil.py:797-851), on demand: returns (count, expo, provider,
    ahead) with count[(gj, gi)] = number of exposures with pixels in the group, expo[(gj, gi)] = their block indices,
    ``provider(keys)`` -> device tensor [sum of the keys' counts, nsamp, nsamp] of sampled PSFs, and ``ahead`` the _HostAhead that
    prepares the host half of coming groups (``ahead.schedule(keys in the order they will be needed)``): the host fetches the
    groups' PSF images at their computation points and evaluates the sampling positions (file broker, WCS: InImage.get_psf_pos,
    outpix2world2inpix), the provider uploads both, and the sampling + cut-out + normalisation (psfutil.py:709-795, 650-656) run
    on the device in one call.  Nothing is read back; a group is fetched when a batch of stamps first needs it (BlockTables).
    ``positions``: "exact" -- the host evaluates ``outpix2world2inpix`` at all nsamp^2 = 146 689 sampling positions per group and
    exposure, as psfutil.py:751-771 does (2.3 MB uploaded each); "lattice" (default) -- at the 17 x 17 Chebyshev-Lobatto nodes spanning
    the same window, and the device forms the positions from them (psfs.lattice_positions: exact for maps of degree < 17 per axis; the
    real chain's deviation from that over the 5" window is below rounding -- tests/test_gpu_refblock.py bounds it at 1e-10 samples on
    affine + distorted maps)."""
    import torch

    from . import psfs

    ns = int(psfgrp.nsamp)
    nst = int(blk.cfg.n1P) + 2
    lin = np.arange(ns) - (ns - 1) / 2.0
    gx, gy = np.meshgrid(lin, lin)
    xy = np.stack([gx.ravel(), gy.ravel()], axis=1) * float(psfgrp.dscale)  # psfutil.py:751-771
    lattice = positions == "lattice"
    if positions not in ("lattice", "exact"):
        raise ValueError(f"positions = {positions!r}")
    if lattice:
        nodes, W_lat = psfs.lattice_nodes_and_weights(lin * float(psfgrp.dscale), LATTICE)
        lx, ly = np.meshgrid(nodes, nodes)  # [a][b]: node a along y, node b along x
        xy = np.stack([lx.ravel(), ly.ravel()], axis=1)
    npos = LATTICE if lattice else ns  # positions per axis the host evaluates
    count, expo = {}, {}
    for gj in range(nst // 2):
        for gi in range(nst // 2):
            used = np.zeros(int(blk.n_inimage), bool)
            for dj in (0, 1):
                for di in (0, 1):
                    st = blk.instamps[2 * gj + dj][2 * gi + di]
                    cnt = getattr(st, "pix_count", None)
                    used |= (np.diff(st.pix_cumsum) if cnt is None else np.asarray(cnt)).astype(bool)
            expos = [int(e) for e in np.flatnonzero(used)]
            if expos:
                count[(gj, gi)], expo[(gj, gi)] = len(expos), expos
    circ, norm = bool(blk.cfg.psf_circ), bool(blk.cfg.psf_norm)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).pin_memory().to(device, non_blocking=True)  # noqa: E731
    oversamp = float(psfgrp.oversamp)

    def host_half(key):
        """PSF images and sampling positions of one group: [(image, yxco)] per exposure of the group (host arrays)."""
        gj, gi = key
        p0 = np.array(blk.instamps[2 * gj][2 * gi].psf_compute_point_pix, dtype=np.float64)
        world = blk.outwcs.all_pix2world(np.array([p0]), 0)[0]
        out = []
        for e in expo[key]:
            im = blk.inimages[e]
            img = np.asarray(im.get_psf_pos(world, use_shortrange=True), dtype=np.float64)
            d = (np.asarray(im.outpix2world2inpix(xy + p0)) - np.asarray(im.outpix2world2inpix(p0[None]))) * oversamp
            out.append((img, d))
        # stacked into page-locked memory here, on the worker: the device loop's thread only queues the copies
        if len({img.shape for img, _ in out}) == 1:
            k = len(out)
            imgs = torch.empty((k,) + out[0][0].shape, dtype=torch.float64, pin_memory=True)
            yx = torch.empty((k, 2, npos, npos), dtype=torch.float64, pin_memory=True)
            iv, yv = imgs.numpy(), yx.numpy()
            for q, (img, d) in enumerate(out):
                iv[q] = img
                yv[q, 0], yv[q, 1] = d[:, 1].reshape(npos, npos), d[:, 0].reshape(npos, npos)
            return imgs, yx
        return [(img, np.stack([d[:, 1].reshape(npos, npos), d[:, 0].reshape(npos, npos)])) for img, d in out]

    def full(yx):
        """device positions [k, 2, ns, ns] of what the host evaluated"""
        return psfs.lattice_positions(yx, W_lat, ns, ctx) if lattice else yx

    ahead = _HostAhead(host_half, host_threads)

    def provider(keys):
        got = [ahead.get(key) for key in keys]
        if all(isinstance(g, tuple) for g in got) and len({g[0].shape[1:] for g in got}) == 1:
            # every group arrived as page-locked stacks of one image size: queue the copies, sample everything in one call.  The copies
            # (1.5 GB for the 72 new groups of a pass at six exposures) run on the upload stream, beside the solve the current stream is
            # busy with; queued behind it they were 30 ms per pass with the device's compute units idle.
            from .stamps import upload_stream

            up, cur = upload_stream(device), torch.cuda.current_stream(device)
            with torch.cuda.stream(up):
                im = torch.cat([g[0].to(device, non_blocking=True) for g in got]) if len(got) > 1 else got[0][0].to(device, non_blocking=True)
                yx = torch.cat([g[1].to(device, non_blocking=True) for g in got]) if len(got) > 1 else got[0][1].to(device, non_blocking=True)
            cur.wait_stream(up)
            im.record_stream(cur)
            yx.record_stream(cur)
            return psfs.sample_psf(im, ns, full(yx), circ, norm, ctx)
        imgs, yxco = [], []
        for g in got:
            pairs = [(g[0][q].numpy(), g[1][q].numpy()) for q in range(g[0].shape[0])] if isinstance(g, tuple) else g
            for img, yx in pairs:
                imgs.append(img)
                yxco.append(yx)
        out = torch.empty((len(imgs), ns, ns), dtype=torch.float64, device=device)
        shapes = {}
        for q, im in enumerate(imgs):  # PSF images of one size are sampled together (normally all of them)
            shapes.setdefault(im.shape, []).append(q)
        for idx in shapes.values():
            got = psfs.sample_psf(up(np.stack([imgs[q] for q in idx])), ns, full(up(np.stack([yxco[q] for q in idx]))), circ, norm, ctx)
            if len(shapes) == 1:
                return got
            out[torch.as_tensor(idx, device=device)] = got
        return out

    return count, expo, provider, ahead


def target_psfs(cfg, psfgrp, device, ctx=None):
    """The output PSF group (psfutil.py:898-929): OUTPSF plus the cfg.outpsf_extra entries."""
    import torch

    from . import psfs

    ns = int(psfgrp.nsamp)
    specs = [(cfg.outpsf, cfg.sigmatarget)] + list(zip(getattr(cfg, "outpsf_extra", []) or [], getattr(cfg, "sigmatarget_extra", []) or []))
    imgs = torch.stack([psfs.get_outpsf(o, s, cfg.use_filter, ns, int(psfgrp.oversamp), device=device, ctx=ctx) for o, s in specs])
    return psfs.sample_psf(imgs, ns, None, bool(cfg.psf_circ), bool(cfg.psf_norm), ctx)


_REPAIR_STATE = {}  # per context: the repair record of the last block's first pass (blockrun.RepairRecord), for inspection -- a block's passes
                    # start from ITS first pass's record, never from another block's

def coadd_output_stamps(blk, psfgrp, flat_penalty=None, batch=None, device="cuda:0", stamps=None, finalize=True, table_capacity=None, ctx=None,
                        host_threads=1, positions="lattice"):
    """Run the stamp loop of ``blk`` on the GPU and fill its block maps (module docstring).  The stamps are those of the
    reference's loop: the window ``blk.j_st_min .. j_st_max, i_st_min .. i_st_max`` of Block._handle_postage_pad (coadd.py:1808-1838;
    default: all n1P x n1P) in cells of 2 x 2, stopping after ``blk.nrun`` stamps when the block carries one (cfg.stoptile,
    1840-1842); ``stamps``: an explicit list of (j_st, i_st) instead.  Of the five quality maps only those named in ``cfg.outmaps``
    ("U", "S", "K", "T", "N"; default all) are written to ``blk``, as coadd.py:1979-1990, 2038-2047 allocate and fill them.
    ``finalize=False`` skips the boundary recovery of coadd.py:2163-2181.  ``batch``: stamps per pass (default: sized
    from the device memory and the table arena, blockrun.plan_block); ``table_capacity``: overlap tables kept resident
    (default: the whole block's, or a third of the free device memory); ``ctx``: the library context to run on (default: the
    process-wide one of the device).  The host half of the PSF groups (PSF images, WCS evaluation of the sampling positions) of
    the coming batches is prepared on ``host_threads`` worker thread(s) while the GPU works on the current one (_HostAhead; one
    thread unless the block's ``inimages`` may be entered from several).  ``positions``: "lattice" (default) -- the WCS chain is evaluated
    at 17 x 17 nodes per PSF group and exposure and the device forms the nsamp^2 sampling positions from them; "exact" -- at all nsamp^2,
    as the reference does (input_psf_groups).  Returns the ``BlockMaps``."""
    from .blockrun import coadd_block, plan_block, stamp_groups
    from .select import InStampPool
    from .stamps import BlockTables

    cfg = blk.cfg
    check_supported(cfg)
    if flat_penalty is None:
        flat_penalty = getattr(cfg, "flat_penalty", 0.0)
    scfg = stamp_config(cfg, psfgrp, blk.n_inimage, flat_penalty)
    for k in ("iter_rtol", "iter_max"):
        if hasattr(cfg, k):
            setattr(scfg, k, getattr(cfg, k))
    pool = InStampPool([(st.x_val, st.y_val, st.data, st.pix_cumsum) for row in blk.instamps for st in row], scfg.n_inframe, device=device)
    count, expo, provider, ahead = input_psf_groups(blk, psfgrp, device, ctx, host_threads, positions)
    target = target_psfs(cfg, psfgrp, device, ctx)
    amp = getattr(cfg, "amp_penalty", None)
    amp = None if amp is None or 0.0 in tuple(amp) else (float(amp[0]), float(amp[1]) * float(psfgrp.oversamp))  # psfutil.py:661-671
    # the device's memory divided before the arenas exist: stamps first, the table and spectra arenas get the rest (blockrun.memory_plan)
    spec_cap = None
    if table_capacity is None and count:
        from .blockrun import memory_plan

        mp_ = memory_plan(scfg, pool, int(cfg.n1P), max(count.values()), n_out=int(target.shape[0]), nfft=int(psfgrp.nfft), ctx=ctx)
        table_capacity, spec_cap = mp_["capacity"], mp_["spec_capacity"]
    tables = BlockTables({k: None for k in count}, target, int(psfgrp.nfft), group_expo=expo, group_count=count, bulk_provider=provider,
                         capacity=None if table_capacity is None else int(table_capacity), spec_capacity=spec_cap, amp_penalty=amp, device=device, ctx=ctx,
                         cells=True,  # groups of 2 x 2 InStamps: cells of the block's grid (coadd.py:207, 329-358)
                         provider_waits=True)  # the provider hands over what the worker threads have prepared
    n1P = int(cfg.n1P)
    window = [int(getattr(blk, k, d)) for k, d in (("j_st_min", 1), ("j_st_max", n1P), ("i_st_min", 1), ("i_st_max", n1P))]
    if stamps is None and (window != [1, n1P, 1, n1P] or getattr(blk, "nrun", None) not in (None, n1P * n1P)):
        from .blockrun import reference_stamp_order

        stamps = reference_stamp_order(*window, nrun=getattr(blk, "nrun", None))
    # the passes first: their order tells which groups' host halves to prepare ahead of the device
    chunks = plan_block(scfg, pool, tables, n1P, batch, stamps=stamps)
    nst = n1P + 2
    per_pass = [list(dict.fromkeys(g for j, i in c for g in stamp_groups(j, i, nst) if g in count)) for c in chunks]
    # the workers may run a whole pass ahead of the device (21 MB of page-locked memory per group at six exposures)
    ahead.ahead = max(ahead.ahead, max((len(p) for p in per_pass), default=0))
    ahead.schedule([g for p in per_pass for g in p])
    try:
        maps = coadd_block(scfg, pool, tables, n1P, int(blk.n_inimage), chunks=chunks, pad_sides=getattr(blk, "pad_sides", "") if finalize else None,
                           postage_pad=int(getattr(cfg, "postage_pad", 0)), origin=(window[0], window[2]),
                           repair_state=_REPAIR_STATE.setdefault(id(tables.ctx), {}))  # (a log of the last block's first-pass repair record; nothing is read from it)
    finally:
        ahead.close()
    blk.out_map, blk.T_weightmap = maps.out_map.cpu().numpy(), maps.T_weightmap.cpu().numpy()
    outmaps = getattr(cfg, "outmaps", "USKTN")
    for c, name, key in (("U", "UC_map", "UC"), ("S", "Sigma_map", "Sigma"), ("K", "kappa_map", "kappa"), ("T", "Tsum_map", "Tsum"), ("N", "Neff_map", "Neff")):
        if c in outmaps:
            setattr(blk, name, maps.maps[key].cpu().numpy())
    return maps
