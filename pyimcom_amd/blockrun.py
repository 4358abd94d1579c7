"""One output block end to end on the GPU: the stamp loop of ``Block.coadd_output_stamps`` (reference
src/pyimcom/coadd.py:2003-2084) with every step device resident --

    InStamp pool -> selection (coadd.py:886-977) -> A, B -> LA kernel -> coaddition -> block maps (1939-2001) -> edge recovery

``tables`` is either one ``PSFGroupTables`` (uniform PSFs over the block, what the synthetic workloads use) or a
``BlockTables`` (one PSF group per 2x2 InStamps as in the reference, psfutil.py:1803-1824): then every stamp gets
its own pair maps -- self tables inside a group, cross tables between the up to four groups its nine InStamps
belong to -- and the tables are computed on demand into the arena.
"""

import os

import numpy as np

from .block import BlockMaps
from .select import select_pixels
from ._lib import ImcomError
from .stamps import NB, BatchBuffers, BlockTables, StampBatch, free_device_bytes, h2d


def stamp_neighbours(j_st, i_st, n2, nst):
    """The nine InStamps of OutStamp (j_st, i_st) (coadd.py:853) with the pivots of coadd.py:918-919 (NaN = None)."""
    left, bottom = (i_st - 1) * n2, (j_st - 1) * n2
    right, top = left + n2 - 1, bottom + n2 - 1
    ids, pvx, pvy = np.full(9, -1, np.int32), np.full(9, np.nan), np.full(9, np.nan)
    for idx, (dj, di) in enumerate((dj, di) for dj in (-1, 0, 1) for di in (-1, 0, 1)):
        jj, ii = j_st + dj, i_st + di
        if 0 <= jj < nst and 0 <= ii < nst:
            ids[idx] = jj * nst + ii
        xp = [left - 0.5, None, right + 0.5][di + 1]
        yp = [bottom - 0.5, None, top + 0.5][dj + 1]
        if xp is not None:
            pvx[idx] = xp
        if yp is not None:
            pvy[idx] = yp
    return ids, pvx, pvy


def _neighbours_of(chunk, n2, nst):
    """stamp_neighbours for a list of stamps, all at once (the same arrays, built by broadcasting)."""
    ji = np.asarray(chunk, dtype=np.int64).reshape(-1, 2)
    dj, di = np.repeat([-1, 0, 1], 3), np.tile([-1, 0, 1], 3)
    jj, ii = ji[:, :1] + dj, ji[:, 1:] + di
    ids = np.where((jj >= 0) & (jj < nst) & (ii >= 0) & (ii < nst), jj * nst + ii, -1).astype(np.int32)
    left, bottom = (ji[:, 1:] - 1) * n2, (ji[:, :1] - 1) * n2
    pvx = np.where(di < 0, left - 0.5, np.where(di > 0, left + n2 - 1 + 0.5, np.nan))
    pvy = np.where(dj < 0, bottom - 0.5, np.where(dj > 0, bottom + n2 - 1 + 0.5, np.nan))
    return [(ids[q], pvx[q], pvy[q]) for q in range(len(ji))]


def prepare_batch(cfg, pool, tables, chunk, n1P, n_expo, ldn=None, buffers=None):
    """Selection, table sets, pair maps and the StampBatch of one chunk of output stamps [(j_st, i_st), ...] of a block
    (everything up to StampBatch.build()); ``pool`` / ``tables`` as for coadd_block."""
    nst = n1P + 2
    counts = np.diff(pool.inst_off)
    nb = _neighbours_of(chunk, cfg.n2, nst)
    ids = np.stack([t[0] for t in nb])
    # capacity: every pixel of the nine neighbours at most
    cap = max(int(counts[t[0][t[0] >= 0]].sum()) for t in nb)
    ld = ldn or max(NB, (cap + NB - 1) // NB * NB)
    # first everything that needs no answer from the GPU -- table sets (queued), pair maps (host) -- then the
    # selection, whose pixel counts the host has to wait for
    psf_slot = maps_ = None
    grouped = isinstance(tables, BlockTables)
    if grouped:
        grp = [[(int(k) // nst >> 1, int(k) % nst >> 1) if k >= 0 else None for k in t[0]] for t in nb]
        local = [list(dict.fromkeys(g for g in gs if g is not None)) for gs in grp]  # distinct groups of each stamp
        tables.require([k for gs in local for k in BlockTables.keys_for(gs)])
        memo = {}  # stamps with the same group list share their maps (at most one list per 2 x 2 cell of groups)
        per = [memo[t] if (t := tuple(gs)) in memo else memo.setdefault(t, tables.stamp_maps(gs, cfg.flat_penalty)) for gs in local]
        maps_ = tuple(np.stack([p[q] for p in per]) for q in range(3))
        lg = np.array([[gs.index(g) if g is not None else 0 for g in row] for row, gs in zip(grp, local)], dtype=np.int64)
    # The selection runs on a side stream with a context of its own: the host has to wait for its pixel counts, and on the
    # main stream that wait would last until the previous batch's builders (queued just before this call) have finished.
    import torch

    main = torch.cuda.current_stream(pool.device)
    side, side_ctx = _side_stream(pool.device)
    with torch.cuda.stream(side):
        x, y, indata, expo, cumsum = select_pixels(pool, ids, np.stack([t[1] for t in nb]), np.stack([t[2] for t in nb]), cfg.rho, ld, ctx=side_ctx)
    for t_ in (x, y, indata, expo):  # produced (and complete: the call returns after its status read-back) on the side stream, used on the main one
        t_.record_stream(main)
    n = cumsum[:, 9]
    keep = (n.max() + NB - 1) // NB * NB if n.max() > 0 else NB  # trim the padding to what the batch needs
    nblk = (n.astype(np.int64) + NB - 1) // NB
    if nblk.max() != nblk.min() and os.environ.get("IMCOM_BLOCK_KEEP_ORDER") != "1":  # (the switch: A/B runs and the test of this reordering)
        # A ragged batch (exposure depth varies over the block): deepest stamps first.  The factorisation and the solves place stamp
        # s on XCD s mod 8; in the late block rows only the deep stamps are active, and in coordinate order their count per XCD is
        # binomial -- the launch then lasts as long as the fullest XCD.  (cfg-4: 792 -> 844 stamps/s.  The maps are placed by
        # coordinates: the order of a batch does not show in the result.)
        import torch

        order = np.argsort(-n.astype(np.int64), kind="stable")
        od = h2d(order, x.device, np.int64)
        x, y, expo, indata = x.index_select(0, od), y.index_select(0, od), expo.index_select(0, od), indata.index_select(0, od)
        cumsum, n = cumsum[order], n[order]
        chunk = [chunk[q] for q in order]
        if grouped:
            per, lg = [per[q] for q in order], lg[order]
            maps_ = tuple(m_[order] for m_ in maps_)
    if grouped:
        import torch

        # stamp-local PSF index of every pixel: lut[position of its InStamp's group in the stamp's list, exposure]
        lut = h2d(np.stack([p[3] for p in per]), x.device, np.int64)  # [B, 4, n_blk_expo]
        cs = h2d(cumsum[:, 1:10], x.device, np.int64)
        seg = torch.searchsorted(cs, torch.arange(keep, device=x.device).expand(len(chunk), keep).contiguous(), right=True).clamp_(max=8)
        lgp = h2d(lg, x.device).gather(1, seg)  # group position of every pixel
        flat = lut.reshape(len(chunk), -1).gather(1, lgp * lut.shape[2] + expo[:, :keep].long())
        valid = torch.arange(keep, device=x.device)[None, :] < h2d(n, x.device, np.int64)[:, None]
        bad = ((flat < 0) & valid).any()  # read by check_batch() once the batch has been solved: no host wait here
        # (its way to the host starts now, into page-locked memory: a read-back after the solve would queue behind whatever the
        # next pass's preparation has put on the stream by then)
        bad_host = torch.empty((), dtype=torch.bool).pin_memory()
        bad_host.copy_(bad, non_blocking=True)
        bad = (bad_host, torch.cuda.current_stream(x.device).record_event())
        psf_slot = flat.clamp_(min=0).to(torch.int32)
    x, y, expo, indata = x[:, :keep], y[:, :keep], expo[:, :keep], indata[:, :, :keep]
    perm = None
    if os.environ.get("IMCOM_PIXEL_ORDER", "psf") != "segments":
        # A stamp's pixels by PSF (group by group, exposure by exposure) instead of the reference's nine InStamp segments with the
        # exposures inside each (coadd.py:937, 688-707): a run of one PSF is then ~N / (groups x exposures) pixels long instead of a
        # segment's share of it, and a 16-row tile of the A builder reads ONE overlap table where it used to straddle two or three
        # (inside a block: 152 -> 114 us per cfg-2 stamp, profiles/r04_negative_results.txt items 3, 10, 14).  A, -B/2, T and indata all
        # live in this order; nothing the block hands out depends on it (the maps and images are sums over the pixels), and
        # ``StampBatchResult.T`` gives T back in the reference's order.  IMCOM_PIXEL_ORDER=segments keeps the reference's order (A/B runs).
        import torch

        key = (psf_slot if grouped else expo).long()
        valid = torch.arange(keep, device=x.device)[None, :] < h2d(n, x.device, np.int64)[:, None]
        perm = torch.where(valid, key, torch.full_like(key, 1 << 30)).sort(dim=1, stable=True).indices
        x, y, expo = x.gather(1, perm), y.gather(1, perm), expo.gather(1, perm)
        indata = indata.gather(2, perm[:, None, :].expand(-1, indata.shape[1], -1))
        if grouped:
            psf_slot = psf_slot.gather(1, perm)
    sb = StampBatch.from_device(cfg, tables, n, x, y, expo, indata,
                                [(i - 1) * cfg.n2 - cfg.fade for _, i in chunk], [(j - 1) * cfg.n2 - cfg.fade for j, _ in chunk],
                                n_expo, ctx=tables.ctx, psf_slot=psf_slot, maps=maps_, buffers=buffers)
    sb.perm = perm  # [B, keep]: position k of a stamp holds the pixel that the reference's order has at perm[k] (None: the reference's order)
    sb.chunk = chunk  # the batch's stamps in the order of its rows (a ragged batch is reordered above)
    sb.bad_psf = bad if grouped else None
    return sb


_SIDE, _BUFS = {}, {}


def _batch_buffers(device):
    import torch

    key = torch.device(device).index or 0
    if key not in _BUFS:
        # ONE set of A, -B/2, T for the pass on the device and the one being prepared: coadd_block builds a pass's A and B only after the
        # pass before it has been solved, coadded and added to the block's maps (all on one stream), and preparing a pass (selection, pair
        # maps) writes none of the three.  (Two sets until round 5: at the reference's production shape -- 0.43 GB of A, -B/2, T per
        # stamp -- the idle set was a third of a pass's memory and held the passes at 84 stamps of a possible 168+.)
        b = BatchBuffers(device)
        _BUFS[key] = (b, b)
    return _BUFS[key]


def release_buffers():
    """Give the per-batch arrays kept between blocks back to torch's allocator."""
    _BUFS.clear()


def _side_stream(device):
    """A second stream with a library context of its own (workspace, pinned ring) per device, for work the host waits on."""
    import torch

    from ._lib import Context

    dev = torch.device(device)
    key = dev.index or 0
    if key not in _SIDE:
        # (highest priority: its few small kernels -- the host waits for their counts -- must not queue behind the thousands of workgroups
        # of a solve that the main stream has in flight.  It IS the upload stream of stamps.h2d: the path keeps to two streams beside the
        # caller's own -- the HIP runtime deals a process's streams onto four hardware queues by default, and streams that share a queue
        # wait for each other; with one stream more than queues the Eigen path lost 30 %.)
        from .stamps import upload_stream

        _SIDE[key] = (upload_stream(dev), Context(key))
    return _SIDE[key]


def check_batch(sb):
    """Raise if a pixel of the batch belongs to an exposure its PSF group holds no PSF for (call after the solve: the flag
    is read back from the device)."""
    bad = getattr(sb, "bad_psf", None)
    if isinstance(bad, tuple):  # (page-locked flag, event behind its copy)
        bad[1].synchronize()
        bad = bool(bad[0])
    if bad is not None and bool(bad):
        raise ValueError("a pixel belongs to an exposure its PSF group holds no PSF for (BlockTables.group_expo)")



def choose_batch(n_stamps, ldn, ldm, n_out=1, free_bytes=None, cu_count=256, cap=256, kernel="Cholesky", nv=1, n_inframe=2):
    """Stamps per batch for a block: as many as memory holds (at most ``cap``), and among those the count for which the block's
    passes -- the short last one included -- take the fewest rounds of workgroups.  A solve launch has batch x ldm/128 workgroups for 2 x ``cu_count`` resident ones: 256 cfg-2 stamps
    are exactly 9 rounds, 128 are 4.5 (every launch ends with a half-empty round: cfg-4 ran 694 stamps/s at 128, 793 at 256),
    64 are 2.25.  Per stamp: A, L (ldn^2 each), -B/2, Y (ldn x ldm per target, fp64), T (fp32), the inverted diagonal blocks;
    the other kernels work on copies in the reference's layout (``stamp_bytes``)."""
    hi = min(int(n_stamps), int(cap))
    if free_bytes is not None:
        hi = max_stamps(int(fill_of(kernel) * free_bytes), ldn, ldm, n_out, kernel, nv, n_inframe, cap=hi)
    if hi >= n_stamps:
        return max(int(n_stamps), 1)
    tiles, slots = max(ldm // NB, 1), 2 * cu_count
    rounds = lambda k: -(-k * tiles // slots)  # noqa: E731  rounds of workgroups of a launch over k stamps
    best, best_cost = hi, None
    for b in range(hi, max(hi // 2, 1) - 1, -1):  # not below half of what fits: every pass costs its launches' fixed parts again
        cost = (n_stamps // b) * rounds(b) + (rounds(n_stamps % b) if n_stamps % b else 0)  # all passes of the block, the short last one too
        if best_cost is None or cost < best_cost:
            best, best_cost = b, cost
    return best


def fill_of(kernel=None):
    """Share of the available device memory a plan may fill.  Every byte of a pass is accounted for -- the per-batch arrays, the
    library workspace by the library's own arithmetic (``stamp_bytes``), the selection's outputs, the block maps -- and all of it comes
    from ONE allocator (torch's: the library works in a torch tensor, ``_lib.Context``), so what is left out is only that allocator's
    rounding of large blocks to 2 MB and the kernels' code objects: 3 %.  (Until round 5 the library allocated its workspace beside
    torch's allocator and the plan kept 20-30 % back for what the two withheld from each other.)"""
    return 0.97


PLAN_HYSTERESIS = 256 << 20  # bytes the available memory may drift before a block is planned anew (a quarter of FIXED_RESERVE)
FIXED_RESERVE = 1 << 30  # bytes a plan leaves alone whatever the block: small per-call buffers, pinned staging's device mirrors, code objects


# The workspace is ONE buffer that the calls of a pass use one after the other: it has the size of the largest of them.  Beside the LA
# kernel that is the overlap tables' inverse transform (chunked to 4 GiB of intermediates, csrc/psf_overlap.hip; the forward spectra are
# asked for in chunks of SPECTRA_CHUNK PSFs, 2.4 MB of workspace each: stamps.BlockTables._ensure_spectra).
TABLE_WS_BYTES = (4 << 30) + (16 << 20)


def pass_bytes(nb, ldn, ldm, n_out=1, kernel="Cholesky", resident=2, nv=1, n_inframe=2, table_ws=0):
    """Device bytes of a pass of ``nb`` stamps while an LA kernel runs: one set of the large StampBatch buffers (A, -B/2, T: shared by
    the pass on the device and the one being prepared, ``_batch_buffers``), the selection's outputs and their trimmed copies for
    ``resident`` batches (coadd_block prepares batch k + 1 while batch k is solved), and the library workspace of the batch being
    solved -- for the Cholesky and Eigen kernels by the library's own count (imcom_solve_chol_workspace / imcom_solve_eigen_workspace
    for exactly ``nb`` stamps and ``nv`` kappa nodes)."""
    import ctypes

    from ._lib import check, lib

    nb = max(int(nb), 1)
    base = 8 * (ldn * ldn + n_out * ldn * ldm) + 4 * n_out * ldn * ldm + 64 * ldm * n_out * (2 + n_inframe)  # A, -B/2, T (f32), maps and images
    small = 3 * int(1.15 * ldn) * (20 + 4 * n_inframe) + 16 * ldn  # selection outputs (x, y, expo, indata at the nine InStamps' capacity), copies, PSF slots
    ws = ctypes.c_size_t(0)
    if kernel == "Cholesky":
        check(lib.imcom_solve_chol_workspace(nb, int(ldn), int(ldm), int(ldm), int(nv), ctypes.byref(ws)))
        return nb * (base + resident * small) + max(ws.value, table_ws) + (2 << 20)
    extra = 12 * ldm * ldn  # -B/2 (f64) and T (f32) in the reference's [m][N] layout
    if kernel == "Eigen":  # the resident entry works on the StampBatch layouts themselves: no copies
        check(lib.imcom_solve_eigen_workspace(nb, int(ldn), int(ldm), int(ldm), ctypes.byref(ws)))
        return nb * (base + resident * small) + max(ws.value, table_ws) + (2 << 20)
    if kernel == "Iterative":
        # the blocked CG (csrc/iter_block.hip): every 4 x 4 patch's selection (6 KB), and the dense union sub-matrices of as many patches
        # at a time as fit a fixed share of 8 GiB (imcom_solve_iter)
        patches = nb * (ldm // 16 + 1)
        return nb * (base + resident * small + extra) + max(min(8 << 30, patches * ITER_PATCH_BYTES) + patches * 6200 + nb * ldm * 256 + (64 << 20), table_ws)
    return nb * (base + resident * small + extra) + max(nb * 8 * ldn * ldn, table_ws)


def stamp_bytes(ldn, ldm, n_out=1, kernel="Cholesky", resident=2, nv=1, n_inframe=2, nb=256):
    """``pass_bytes`` per stamp at a pass of ``nb`` stamps (a planner's first estimate; ``max_stamps`` is exact)."""
    return -(-pass_bytes(nb, ldn, ldm, n_out, kernel, resident, nv, n_inframe) // nb)


def max_stamps(avail, ldn, ldm, n_out=1, kernel="Cholesky", nv=1, n_inframe=2, cap=256, table_ws=0):
    """The largest pass (at most ``cap`` stamps, at least one) whose ``pass_bytes`` fit ``avail`` bytes."""
    lo, hi = 1, max(int(cap), 1)
    if pass_bytes(hi, ldn, ldm, n_out, kernel, 2, nv, n_inframe, table_ws) <= avail:
        return hi
    while lo < hi:  # pass_bytes grows with nb: bisection for the last nb that fits
        mid = (lo + hi + 1) // 2
        if pass_bytes(mid, ldn, ldm, n_out, kernel, 2, nv, n_inframe, table_ws) <= avail:
            lo = mid
        else:
            hi = mid - 1
    return lo


def block_maps_bytes(n1P, n2, fade, n_inframe, n_out=1, n_expo=1):
    """Device bytes of a block's BlockMaps (block.py): the maps themselves and, with fade > 0, the four parity layers the stamps'
    tiles are kept in (allocated when the first tiles arrive: after the plan was made)."""
    ns = n1P * n2 + 2 * fade
    maps = n_out * ns * ns * 4 * (n_inframe + 5) + n_out * n_expo * n1P * n1P * 4
    layers = n_out * 4 * ns * ns * (4 * n_inframe + 4 * 3 + 8 * 2) if fade > 0 else 0  # out_map / UC, Sigma, kappa (f32); Tsum, Neff (f64)
    return maps + layers


def available_bytes(device, ctx=None):
    """Device memory a pass can draw on: what the driver reports free, what torch's allocator holds without using it, and what earlier
    passes left in the context's workspace and in the per-batch buffers (both are torch tensors that the next pass reuses or replaces)."""
    import torch

    free = free_device_bytes(device)
    if ctx is not None and getattr(ctx, "_ws", None) is not None:
        free += int(ctx._ws.numel())
    elif ctx is not None and hasattr(ctx, "workspace_bytes") and not getattr(ctx, "_owns_ws", False):
        free += int(ctx.workspace_bytes())
    key_ = torch.device(device).index or 0
    if key_ in _BUFS:
        free += _BUFS[key_][0].nbytes()
    return free


def tile_tables(h, w, n, n_out):
    """Overlap tables a tile of h x w cells of 2 x 2 stamps needs resident (the cost model of plan_batches): self + input-output sets of its
    (h + 1)(w + 1) groups, cross sets of the neighbouring pairs."""
    S = n * (n + 1) // 2 + n_out * n
    return (h + 1) * (w + 1) * S + ((h + 1) * w + h * (w + 1) + 2 * h * w) * n * n


def memory_plan(cfg, pool, n1P, n_psf, n_out=1, nfft=None, ctx=None, cap=256, kernel=None, nv=None):
    """How a block's device memory is divided BEFORE its table arenas exist: {"capacity": overlap tables, "spec_capacity": spectra rows,
    "stamps": stamps per pass the rest holds}, to be handed to ``BlockTables(capacity=..., spec_capacity=...)``.  Stamps first: a pass is
    as large as memory allows up to ``cap`` provided the arenas still hold two passes' worth of tables and spectra (the pass on the device
    and the one being prepared); what is left beyond that goes to the arenas, up to the whole block (nothing is then computed twice).
    At cfg-2 size (0.15 GB per stamp) that is passes of 256 and every table of a 48 x 48 block resident; at the reference's production
    shape (1.2 GB per stamp: paper4) it is passes of ~160 and arenas of two passes -- where the fixed thirds of BlockTables' defaults
    left the stamps passes of 32-40."""
    import ctypes

    import torch

    from ._lib import lib

    kernel = kernel or cfg.kernel
    nv = nv or len(cfg.kappaC)
    nfft = nfft or cfg.nfft
    win, exact = stamp_pixels(cfg, pool, n1P)
    nmax = int(win.max()) if exact else int(1.1 * win.max()) + 64
    ldn = max(NB, (nmax + NB - 1) // NB * NB)
    ldm = (cfg.m + NB - 1) // NB * NB
    avail = int(fill_of(kernel) * available_bytes(pool.device, ctx)) - FIXED_RESERVE - block_maps_bytes(n1P, cfg.n2, cfg.fade, cfg.n_inframe, n_out, n_psf)
    tb = 8 * (cfg.nsamp + 12) ** 2
    sb = 8 * int(lib.imcom_psf_spectra_size(cfg.nsamp, nfft)) or 1
    ng = (n1P + 3) // 2
    S = n_psf * (n_psf + 1) // 2 + n_out * n_psf
    block_tables = ng * ng * S + (2 * ng * (ng - 1) + 2 * (ng - 1) ** 2) * n_psf * n_psf
    block_rows = n_out + ng * ng * n_psf
    pb = lambda k: pass_bytes(k, ldn, ldm, n_out, kernel, 2, nv, cfg.n_inframe, TABLE_WS_BYTES)  # noqa: E731
    c = min(cap, n1P * n1P)
    while True:
        side = max(1, int(np.ceil(np.sqrt(max(c, 4) / 4.0))))
        need_t = min(block_tables, 2 * tile_tables(side, side, n_psf, n_out))
        need_r = min(block_rows, n_out + 2 * (side + 1) ** 2 * n_psf)
        if pb(c) + need_t * tb + need_r * sb <= avail or c <= 4:
            break
        c = max(4, c - 4)
    per = pb(c) // c
    rest = max(avail - pb(c), need_t * tb + need_r * sb)
    cap_t = int(min(block_tables, max(need_t, (rest - need_r * sb) * 0.85 // tb)))
    cap_r = int(min(block_rows, max(need_r, (rest - cap_t * tb) // sb)))
    return {"capacity": max(cap_t, 1), "spec_capacity": max(cap_r, n_out + 4 * n_psf), "stamps": int(c), "ldn": ldn, "bytes_per_stamp": int(per),
            "available": int(avail), "block_tables": int(block_tables)}


ITER_PATCH_BYTES = 1024 * 1024 * 8 + 3 * 1024 * 16 * 8  # iter_block_patch_bytes(1024) (csrc/iter_block.hip): sub-matrix, right-hand sides, x, q of a 4 x 4 patch


def stamp_groups(j_st, i_st, nst):
    """The PSF groups (2 x 2 InStamps, SysMatA.ji_st2psf psfutil.py:1803-1824) the nine InStamps of OutStamp (j_st, i_st)
    belong to, in the order they are first met (coadd.py:853)."""
    return tuple(dict.fromkeys((jj >> 1, ii >> 1) for jj in (j_st - 1, j_st, j_st + 1) for ii in (i_st - 1, i_st, i_st + 1)
                               if 0 <= jj < nst and 0 <= ii < nst))


def chunk_keys(chunk, nst):
    """Table sets a chunk of stamps needs resident together."""
    seen = dict.fromkeys(stamp_groups(j, i, nst) for j, i in chunk)
    return list(dict.fromkeys(k for gs in seen for k in BlockTables.keys_for(gs)))


def plan_batches(todo, nst, cap, tables=None, tiles_per_stamp=18, stamp_cost=0.6e-3, table_cost=2.3e-6, slots=512):
    """Cut the stamps ``todo`` [(j_st, i_st), ...] of a block into batches of at most ``cap`` stamps.

    With one PSF group for the whole block the order is kept.  With a group per 2 x 2 InStamps (``tables``: BlockTables) the
    batches are rectangular TILES of cells of 2 x 2 stamps -- the four stamps of a cell draw on the same four groups, and a
    tile of h x w cells touches (h + 1)(w + 1) groups where a row of the block 84 stamps wide would touch 2 x 43 for the same
    number of stamps -- chosen among the balanced tilings of the block by a cost model: rounds of workgroups of the batch's
    launches (``tiles_per_stamp`` workgroups per stamp, ``slots`` resident at once, ``stamp_cost`` seconds per stamp when
    the rounds are full) plus ``table_cost`` per overlap table the tile needs.  Tiles whose table sets exceed the arena
    (``tables.capacity``) are halved until they fit.  Tiles are visited boustrophedon, so that consecutive batches share a
    column of groups whose sets are still resident."""
    todo = list(todo)
    if not todo:
        return []
    cap = max(1, int(cap))
    if tables is None:
        return [todo[c0 : c0 + cap] for c0 in range(0, len(todo), cap)]
    cj = np.array([(j + 1) >> 1 for j, _ in todo])
    ci = np.array([(i + 1) >> 1 for _, i in todo])
    j0, i0 = int(cj.min()), int(ci.min())
    H, W = int(cj.max()) - j0 + 1, int(ci.max()) - i0 + 1  # cells
    n, S = tables.n_max, tables.n_max * (tables.n_max + 1) // 2 + tables.n_out * tables.n_max
    per_round = stamp_cost * slots / tiles_per_stamp  # seconds per full round of workgroups
    split = lambda L, k: [L * q // k for q in range(k + 1)]  # noqa: E731  balanced cut of L cells into k runs

    def cost(nr, nc):
        h, w = np.diff(split(H, nr)).astype(np.float64), np.diff(split(W, nc)).astype(np.float64)
        st = 4.0 * np.outer(h, w)
        if st.max() > cap or (st.max() < cap / 4 and nr * nc > 1):
            return None  # a tile above the batch limit / a tiling far finer than needed
        tabs = np.outer(h + 1, w + 1) * S + (np.outer(h + 1, w) + np.outer(h, w + 1) + 2.0 * np.outer(h, w)) * (n * n)
        return float((np.ceil(st * tiles_per_stamp / slots) * per_round + tabs * table_cost).sum())

    best = None
    for nr in range(1, H + 1):
        for nc in range(1, W + 1):
            c = cost(nr, nc)
            if c is not None and (best is None or c < best[0] - 1e-12):
                best = (c, nr, nc)
    _, nr, nc = best if best is not None else (0.0, H, W)
    eh, ew = np.array(split(H, nr)), np.array(split(W, nc))
    tj = np.searchsorted(eh, cj - j0, side="right") - 1
    ti = np.searchsorted(ew, ci - i0, side="right") - 1
    bins = {}
    for q, key in enumerate(zip(tj.tolist(), ti.tolist())):
        bins.setdefault(key, []).append(todo[q])
    out = []

    cell_of = lambda t: ((t[0] + 1) >> 1, (t[1] + 1) >> 1)  # noqa: E731

    def emit(first):
        # cells stay together and in order (the stamps of a cell share their pair maps and their tables in cache).  (A loop with its own
        # stack, not a recursive closure: a function that refers to itself is a reference cycle, and this one holds ``tables`` -- a
        # block's table arena, tens of GB, stayed alive until the next pass of the garbage collector.)
        stack = [first]
        while stack:
            chunk = stack.pop()
            chunk.sort(key=lambda t: (cell_of(t), t))
            if len(chunk) > cap or (len(chunk) > 1 and tables.demand(chunk_keys(chunk, nst)) > tables.capacity):
                cells = list(dict.fromkeys(cell_of(t) for t in chunk))
                if len(cells) > 1:  # halve along the longer side of the tile, on a cell boundary
                    js, is_ = sorted({c[0] for c in cells}), sorted({c[1] for c in cells})
                    if len(js) >= len(is_):
                        head = set(js[: len(js) // 2])
                        parts = [t for t in chunk if cell_of(t)[0] in head], [t for t in chunk if cell_of(t)[0] not in head]
                    else:
                        head = set(is_[: len(is_) // 2])
                        parts = [t for t in chunk if cell_of(t)[1] in head], [t for t in chunk if cell_of(t)[1] not in head]
                else:
                    parts = chunk[: len(chunk) // 2], chunk[len(chunk) // 2 :]
                stack.append(list(parts[1]))  # (the first half is taken off the stack first: the order of the recursive form)
                stack.append(list(parts[0]))
            else:
                out.append(chunk)

    for a in range(nr):
        for b in (range(nc) if a % 2 == 0 else range(nc - 1, -1, -1)):
            if (a, b) in bins:
                emit(bins[(a, b)])
    return out


def estimate_pixels(cfg, pool, n1P):
    """Pixels every stamp of a block will select (coadd.py:886-977), [n1P, n1P] float, estimated from its nine InStamps' counts:
    the centre cell whole, of the four edge cells the strip within rho (a fraction rho / n2 of their area), of the corner cells
    a quarter disc.  Batches are sized with it (+ 10 %); the leading dimension a batch really gets comes from the counts the
    selection kernel returns (prepare_batch)."""
    nst = n1P + 2
    counts = np.diff(pool.inst_off).reshape(nst, nst).astype(np.float64)
    fe = min(1.0, cfg.rho / cfg.n2)
    frac = {0: 1.0, 1: fe, 2: min(1.0, np.pi / 4.0 * fe * fe)}
    return sum(frac[(dj != 1) + (di != 1)] * counts[dj : dj + n1P, di : di + n1P] for dj in range(3) for di in range(3))


def reference_stamp_order(j_st_min, j_st_max, i_st_min, i_st_max, nrun=None):
    """The (j_st, i_st) the reference's stamp loop visits, in its order (coadd.py:2056-2064): cells of 2 x 2 stamps, row by row of
    cells, the four stamps of a cell in product(range(2), range(2)) order, stopping after ``nrun`` stamps (cfg.stoptile,
    coadd.py:1840-1842).  The window comes from Block._handle_postage_pad (1808-1838)."""
    if (j_st_max + 1 - j_st_min) % 2 or (i_st_max + 1 - i_st_min) % 2:
        raise ValueError(f"Size must be even: y={j_st_min}..{j_st_max}, x={i_st_min}..{i_st_max}")  # coadd.py:2052-2055
    out = []
    for j in range(j_st_min, j_st_max + 1, 2):
        for i in range(i_st_min, i_st_max + 1, 2):
            for dj in range(2):
                for di in range(2):
                    out.append((j + dj, i + di))
                    if nrun is not None and len(out) == nrun:
                        return out
    return out


def count_pixels(cfg, pool, n1P):
    """Pixels every stamp of a block selects (coadd.py:886-977), [n1P, n1P] int, EXACT: the selection kernel itself run over the
    whole block, a few hundred stamps per call, only its counts read back (a few milliseconds per block; cached on the pool).
    With exact counts the passes are sized without the 10 % margin an estimate needs -- at cfg-3 size the margin alone turned
    passes of 256 stamps into passes of 128 (ldn 3328 instead of 3072 in the memory model).  Pools that do not live on a GPU
    (the planner's host tests) have no exact counts: returns None."""
    if getattr(pool, "device", None) is None or getattr(pool.device, "type", "") != "cuda" or not hasattr(pool, "x"):
        return None
    key = (float(cfg.rho), int(cfg.n2), int(n1P))
    cache = getattr(pool, "_pixel_counts", None)
    if cache is not None and cache[0] == key:
        return cache[1]
    nst = n1P + 2
    todo = [(j, i) for j in range(1, n1P + 1) for i in range(1, n1P + 1)]
    per = np.diff(pool.inst_off)
    out = np.zeros((n1P, n1P), np.int64)
    step = 512
    for c0 in range(0, len(todo), step):
        nb = _neighbours_of(todo[c0 : c0 + step], cfg.n2, nst)
        cap = max(int(per[t[0][t[0] >= 0]].sum()) for t in nb)
        _, _, _, _, cumsum = select_pixels(pool, np.stack([t[0] for t in nb]), np.stack([t[1] for t in nb]), np.stack([t[2] for t in nb]), cfg.rho,
                                           max(NB, (cap + NB - 1) // NB * NB))
        for (j, i), n in zip(todo[c0 : c0 + step], cumsum[:, 9]):
            out[j - 1, i - 1] = int(n)
    pool._pixel_counts = (key, out)
    return out


def stamp_pixels(cfg, pool, n1P):
    """(pixels per stamp [n1P, n1P], exact?): the selection's own counts when the pool is on a GPU, else the estimate."""
    exact = count_pixels(cfg, pool, n1P)
    return (exact.astype(np.float64), True) if exact is not None else (estimate_pixels(cfg, pool, n1P), False)


def plan_block(cfg, pool, tables, n1P, batch=None, ldn=None, stamps=None):
    """The batches coadd_block runs: list of lists of (j_st, i_st).  ``batch=None``: sized from the block's largest stamp, the
    free device memory and -- with a BlockTables -- the table arena (``choose_batch`` / ``plan_batches``); an explicit ``batch``
    cuts the visiting order every ``batch`` stamps (row by row as coadd.py:2049-2052; cell by cell with a BlockTables)."""
    nst = n1P + 2
    todo = [(j, i) for j in range(1, n1P + 1) for i in range(1, n1P + 1)] if stamps is None else [(int(j), int(i)) for j, i in stamps]
    if not todo:
        return []
    grouped = isinstance(tables, BlockTables)
    win, exact = stamp_pixels(cfg, pool, n1P)
    cap_pix = int(max(win[j - 1, i - 1] for j, i in todo)) if exact else int(1.1 * max(win[j - 1, i - 1] for j, i in todo)) + 64
    ldn_max = ldn or max(NB, (cap_pix + NB - 1) // NB * NB)
    ldm = (cfg.m + NB - 1) // NB * NB
    if batch is None:
        # what a pass can draw on: free memory, and what earlier passes / blocks left in the library workspace and in the per-batch
        # buffers -- both are reused (counting them as taken made a second block of the same kind plan passes half the size)
        n_out = int(getattr(tables, "n_out", 1))
        nv_, nf_ = len(cfg.kappaC), cfg.n_inframe
        free = available_bytes(pool.device, getattr(tables, "ctx", None)) - int((FIXED_RESERVE + block_maps_bytes(n1P, cfg.n2, cfg.fade, nf_, n_out, getattr(tables, "n_psf", 1)))
                                                                                   / fill_of(cfg.kernel))
        free = max(free, 0)
        # A plan is kept while the memory it was made for has not moved by more than PLAN_HYSTERESIS: small buffers come and go between
        # two plans of the same block (selection outputs, a finished block's maps, the allocator's rounding -- tens of MB), and a plan that
        # sits on the border between k and k + 1 stamps per pass must not flip with them
        key_ = (cfg.name, cfg.kernel, nv_, nf_, n_out, n1P, len(todo), ldn_max, ldm, grouped)
        memo = getattr(tables, "_plan_memo", None)
        if memo is not None and memo[0] == key_ and abs(memo[1] - free) <= PLAN_HYSTERESIS:
            free = memo[1]
        else:
            tables._plan_memo = (key_, free)
        if grouped:  # tiles of 2 x 2-stamp cells, sized by memory and by the table arena (plan_batches)
            cap = max_stamps(int(fill_of(cfg.kernel) * free), ldn_max, ldm, n_out, cfg.kernel, nv_, nf_, cap=256, table_ws=TABLE_WS_BYTES)
        else:
            cap = choose_batch(len(todo), ldn_max, ldm, n_out, free, kernel=cfg.kernel, nv=nv_, n_inframe=nf_)
    else:
        cap = max(1, int(batch))
    if not grouped:
        return [todo[c0 : c0 + cap] for c0 in range(0, len(todo), cap)]
    # The four stamps (2a-1..2a, 2b-1..2b) of a cell draw on the same four PSF groups (their InStamps 2a-2..2a+1 are group
    # rows a-1, a): a batch is a tile of cells, visited cell by cell (the maps are placed by coordinates, so the order of the
    # visit does not show in the result).  An explicit ``batch`` keeps the cell order and cuts it every ``batch`` stamps.
    if batch is None:
        nbar = float(np.mean([win[j - 1, i - 1] for j, i in todo]))
        per_stamp = (2.0 * nbar * nbar * cfg.m + nbar**3 / 3.0 + 165.0 * nbar * nbar) / 45e12  # seconds at the rate the path reaches
        return plan_batches(todo, nst, cap, tables, tiles_per_stamp=max(ldm // NB, 1), stamp_cost=per_stamp)
    todo.sort(key=lambda t: ((t[0] + 1) >> 1, (t[1] + 1) >> 1, t[0], t[1]))
    chunks = []
    for c0 in range(0, len(todo), cap):
        part = [todo[c0 : c0 + cap]]
        while part:  # a chunk whose table sets exceed the arena is halved
            c = part.pop(0)
            if len(c) > 1 and tables.demand(chunk_keys(c, nst)) > tables.capacity:
                part[:0] = [c[: len(c) // 2], c[len(c) // 2 :]]
            else:
                chunks.append(c)
    return chunks


class RepairRecord:
    """What a block's FIRST pass saw of _cholesky_wrapper's repair (lakernel.py:262-279) -- the share of its stamps that took it and the
    largest |w[0]| among them -- kept for the block's other passes: each of them tells the library to expect the repair and where the
    smallest-eigenvalue iterations start (imcom_ctx_set_repair_hint) from THIS record, whoever runs the pass and whatever ran before it
    in the process.  The first pass itself starts blind.  So the inputs of every pass are a function of the block alone, and a block
    whose passes are shared between processes (pyimcom_amd.farm keeps the record in a file beside the block's claims) has the bits of
    the single process's block -- until round 6 a pass started from the pass before it, and a repaired stamp's w[0] depended on the
    partition at the 1e-13 level.  In memory here; ``get`` waits for ``put`` only in the file-backed form."""

    def __init__(self):
        self.value = None

    def put(self, share, hint):
        self.value = {"share": float(share), "hint": None if hint is None else float(hint)}

    def get(self):
        return self.value


def coadd_block(cfg, pool, tables, n1P, n_expo, batch=None, pad_sides="", postage_pad=0, ldn=None, pipeline=True, stamps=None, chunks=None,
                claim=None, origin=(1, 1), repair_state=None, repair_record=None, first_chunk=0):
    """Coadd the n1P x n1P output stamps of a block.  ``pool``: InStampPool of the (n1P+2)^2 InStamps in row-major
    order (index j * nst + i, coadd.py:207); ``tables``: PSFGroupTables or BlockTables.  ``stamps``: the (j_st, i_st) to
    coadd (default all n1P x n1P); ``pad_sides=None`` leaves the boundary recovery of coadd.py:2163-2181 out.  ``batch``:
    stamps per pass, or ``chunks``: the passes themselves (default: ``plan_block``).  ``claim(q) -> bool`` (optional) is asked
    right before pass q is prepared; a pass it refuses is left out (another process coadds it: pyimcom_amd.farm shares a block's
    passes between the GPUs of a node) -- the passes that were run are listed in ``maps.chunks_done``.  ``origin``: (j_st_min,
    i_st_min) of the reference's loop (coadd.py:1808-1838): with fade > 0 overlapping stamps are summed in the order of that loop
    whatever the batches are (block.py).  ``repair_record`` (a RepairRecord; default: one of this call's own) and ``first_chunk`` (the
    index in ``chunks`` of the block's first pass): the Cholesky repair's expectation and starting shift of every pass but the first come
    from the block's first pass (RepairRecord).  ``repair_state``: a dict that receives {"share", "hint"} of that record (for a driver's
    log; until round 6 it carried them from block to block -- a block's result then depended on the blocks before it).  Returns the
    BlockMaps."""
    nst = n1P + 2
    assert pool.n_inst == nst * nst
    maps = BlockMaps(n1P, cfg.n2, cfg.fade, cfg.n_inframe, n_expo, ctx=tables.ctx, device=str(pool.device),
                     n_out=int(getattr(tables, "n_out", 1)), origin=origin)
    if chunks is None:
        if isinstance(tables, BlockTables) and claim is None and tables.eager_groups:
            # every PSF group of the stamps to come is sampled / transformed sooner or later: queue as many as the spectra arena holds
            # NOW, so that the device works while the host plans the passes (12 ms at n1P = 48) and prepares the first one
            # (BlockTables(eager_groups=True): the provider only queues device work)
            if stamps is None:
                tables.prefetch(list(tables.psf))  # the whole block: every group
            else:
                # (a stamp's nine InStamps j-1 .. j+1 lie in the group rows (j-1) >> 1 and (j+1) >> 1, likewise the columns: stamp_groups)
                ji = np.asarray(stamps).reshape(-1, 2)
                gs = np.unique(np.concatenate([np.stack([(ji[:, 0] + a) >> 1, (ji[:, 1] + b) >> 1], axis=1) for a in (-1, 1) for b in (-1, 1)]), axis=0)
                tables.prefetch([(int(a), int(b)) for a, b in gs])
        chunks = plan_block(cfg, pool, tables, n1P, batch, ldn, stamps)
    chunks = [[(int(j), int(i)) for j, i in c] for c in chunks]
    maps.chunks_done, maps.chunk_sizes, maps.info_nonzero = [], [len(c) for c in chunks], 0
    if isinstance(tables, BlockTables) and chunks and claim is None:
        # the PSF groups of the first batches are sampled / transformed before the host turns to the per-stamp bookkeeping (a provider that
        # waits for host work is only asked for the first pass: with two, the device idled until the second pass's groups were there)
        first = chunks[:1] if tables.provider_waits else chunks[:2]
        tables.prefetch(dict.fromkeys(g for c in first for t in c for g in stamp_groups(t[0], t[1], nst)))
    # the large per-batch arrays (one set: see _batch_buffers), kept from block to block
    bufs = _batch_buffers(pool.device)
    if chunks and ldn is None:
        # sized once for the block's largest batch (estimated, as the plan's): a buffer that has to grow in the middle of the
        # block costs a device allocation of ~10 GB, 0.1 s with the GPU idle
        win, exact = stamp_pixels(cfg, pool, n1P)
        nmax_ = max(win[j - 1, i - 1] for c in chunks for j, i in c)
        ld_pre = (int(nmax_ if exact else 1.05 * nmax_ + 32) + NB - 1) // NB * NB
        bmax, ldm, O = max(len(c) for c in chunks), (cfg.m + NB - 1) // NB * NB, int(getattr(tables, "n_out", 1))
        import torch

        bufs[0].take("A", (bmax, ld_pre, ld_pre), torch.float64)
        bufs[0].take("Bt", (O, bmax, ld_pre, ldm), torch.float64)
        bufs[0].take("Tt", (O, bmax, ld_pre, ldm), torch.float32)
    todo = iter(range(len(chunks)))
    count = [0]
    claimed = []      # the passes this process took (claim(q) said yes): every one of them must end up in maps.chunks_done
    pending = [None]  # a pass that was taken but whose preparation failed: the next call prepares THAT pass again, never a new one

    def next_batch():
        """The next pass this process may run, prepared.  Taking a pass (the claim, under the farm a file other ranks see) and preparing it
        are separate: when the preparation raises -- out of memory in its gathers, IMCOM_ERR_NOMEM of a table call -- the pass stays
        ``pending`` and the next call prepares it again, so a pass that was claimed is never skipped."""
        q = pending[0]
        if q is None:
            for q_ in todo:
                if chunks[q_] and (claim is None or claim(q_)):
                    q = q_
                    claimed.append(q)
                    break
        if q is None:
            return None
        pending[0] = q
        sb = prepare_batch(cfg, pool, tables, chunks[q], n1P, n_expo, ldn, buffers=bufs[count[0] & 1])
        pending[0] = None
        sb.chunk_index, sb.buf_index = q, count[0] & 1
        count[0] += 1
        return sb

    # software pipeline: the next chunk is prepared on the host (and its selection / table kernels queued) right after the current
    # one's A and B builds AND its solve have been queued (solve_begin: the Cholesky kernel's launches without their read-back), i.e.
    # while the GPU is busy with them; only then does the host block in the solve's status read-back (solve_end).  (Until round 4 the
    # solve was queued after the next chunk's preparation: where that takes longer than the builds -- the Block seam, whose provider
    # hands over host arrays -- the device idled for the difference in every pass.)
    def out_of_memory(e):
        return e.status == -3 or (e.status == -2 and "out of memory" in str(e))  # IMCOM_ERR_NOMEM, or a launch / allocation HIP refused for lack of memory

    def in_halves(sb):
        """The emergency route.  A plan fills 0.97 of what torch's allocator reports as available with exact figures for this process
        (plan_block), but not for what else shares the device: a pass the device has no memory for after all is run again as two passes
        of half the stamps, cut on a cell boundary, in the failed pass's buffers.  The failure may have come from solve_begin, from the
        preparation of the NEXT pass or from solve_end, so launches of the failed pass (its builds, a deferred solve) may still be queued
        on the ONE set of A / -B/2 / T arrays the halves are about to overwrite: the synchronize below is what makes that safe -- nothing
        of the failed pass is read after it, its results are discarded."""
        import torch

        torch.cuda.synchronize()
        tables.ctx.release_workspace()
        torch.cuda.empty_cache()
        chunk, h = list(sb.chunk), max(4, len(sb.chunk) // 2 // 4 * 4)
        done = []
        for part in (chunk[:h], chunk[h:]):
            if not part:
                continue
            sp = prepare_batch(cfg, pool, tables, part, n1P, n_expo, ldn, buffers=bufs[sb.buf_index])
            sp.build()
            sp.solve()
            check_batch(sp)
            sp.coadd()
            maps.add(sp.results(), [j for j, _ in sp.chunk], [i for _, i in sp.chunk])
            done.append(len(part))
        return done

    nxt = next_batch()
    maps.passes_halved = 0
    maps.pass_seconds = []  # host clock per pass (a pass ends with the solve's status read-back: what is still queued behind it belongs to the next)
    import time

    t_pass = time.perf_counter()
    record = repair_record if repair_record is not None else RepairRecord()

    def pass_inputs(q):
        """(expect the repair, where its iterations start) for pass q: nothing for the block's first pass, the first pass's record for the others"""
        if q == first_chunk:
            return False, None
        rec = record.get()  # (file-backed: waits until the process that runs the first pass has written it)
        if rec is None:  # no first pass in this call's plan (a driver that runs a block's later passes alone): blind
            return False, None
        return rec["share"] >= StampBatch.EXPECT_REPAIR, rec["hint"]

    while nxt is not None:
        sb = nxt
        sb.build()
        try:
            expect, hint = pass_inputs(sb.chunk_index)
            sb.solve_begin(expect_repair=expect, repair_hint=hint)
            if pipeline:
                nxt = next_batch()
            sb.solve_end()
        except (ImcomError, RuntimeError) as e:
            # the emergency route (a plan is exact about this process's memory, not about what else shares the device): the pass again as two
            oom = out_of_memory(e) if isinstance(e, ImcomError) else "out of memory" in str(e).lower()
            if not oom or len(sb.chunk) < 8:
                raise
            prepared = nxt if (pipeline and nxt is not sb) else None
            in_halves(sb)
            maps.passes_halved += 1
            maps.chunks_done.append(sb.chunk_index)
            if sb.chunk_index == first_chunk:  # (its halves were solved blind by the synchronous entry: the record says "no expectation, no hint")
                record.put(0.0, None)
            if prepared is not None:
                # the halves' table sets may have replaced the ones the already prepared next pass was given (its slots and maps were fixed
                # when it was prepared): prepare it again
                q_, b_ = prepared.chunk_index, prepared.buf_index
                nxt = prepare_batch(cfg, pool, tables, chunks[q_], n1P, n_expo, ldn, buffers=bufs[b_])
                nxt.chunk_index, nxt.buf_index = q_, b_
            elif not pipeline or nxt is sb:
                nxt = next_batch()  # (a pass whose preparation failed above is still pending: it is prepared again, not skipped)
            continue
        if sb.chunk_index == first_chunk:
            record.put(float(getattr(sb, "repair_share", 0.0)), getattr(sb, "repair_absmax", None))
        check_batch(sb)
        sb.coadd()
        if not pipeline:
            nxt = next_batch()
        maps.add(sb.results(), [j for j, _ in sb.chunk], [i for _, i in sb.chunk])
        maps.chunks_done.append(sb.chunk_index)
        maps.info_nonzero += int(sum(int((np.asarray(i_) != 0).sum()) for i_ in sb.info_o))  # stamps repaired (Cholesky) / re-solved in the eigenbasis (Eigen)
        maps.pass_seconds.append(time.perf_counter() - t_pass)
        t_pass = time.perf_counter()
    if repair_state is not None and record.value is not None:
        repair_state.update(record.value)
    if sorted(claimed) != sorted(maps.chunks_done):  # (never seen; a pass that was taken and not coadded would be a silent hole in the maps)
        raise RuntimeError(f"coadd_block: passes taken {sorted(claimed)} but coadded {sorted(maps.chunks_done)}")
    if pad_sides is not None:
        maps.finalize(pad_sides, postage_pad)
    return maps
