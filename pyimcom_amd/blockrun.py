"""One output block end to end on the GPU: the stamp loop of ``Block.coadd_output_stamps`` (reference
src/pyimcom/coadd.py:2003-2084) with every step device resident --

    InStamp pool -> selection (coadd.py:886-977) -> A, B -> LA kernel -> coaddition -> block maps (1939-2001) -> edge recovery

``tables`` is either one ``PSFGroupTables`` (uniform PSFs over the block, what the synthetic workloads use) or a
``BlockTables`` (one PSF group per 2x2 InStamps as in the reference, psfutil.py:1803-1824): then every stamp gets
its own pair maps -- self tables inside a group, cross tables between the up to four groups its nine InStamps
belong to -- and the tables are computed on demand into the arena.
"""

import numpy as np

from .block import BlockMaps
from .select import select_pixels
from .stamps import NB, BlockTables, StampBatch


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


def prepare_batch(cfg, pool, tables, chunk, n1P, n_expo, ldn=None):
    """Selection, table sets, pair maps and the StampBatch of one chunk of output stamps [(j_st, i_st), ...] of a block
    (everything up to StampBatch.build()); ``pool`` / ``tables`` as for coadd_block."""
    nst = n1P + 2
    counts = np.diff(pool.inst_off)
    nb = [stamp_neighbours(j, i, cfg.n2, nst) for j, i in chunk]
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
        per = [tables.stamp_maps(gs, cfg.flat_penalty) for gs in local]
        maps_ = tuple(np.stack([p[q] for p in per]) for q in range(3))
        lg = np.array([[gs.index(g) if g is not None else 0 for g in row] for row, gs in zip(grp, local)], dtype=np.int64)
    x, y, indata, expo, cumsum = select_pixels(pool, ids, np.stack([t[1] for t in nb]), np.stack([t[2] for t in nb]), cfg.rho, ld,
                                               ctx=tables.ctx)
    n = cumsum[:, 9]
    keep = (n.max() + NB - 1) // NB * NB if n.max() > 0 else NB  # trim the padding to what the batch needs
    if grouped:
        import torch

        # stamp-local PSF index of every pixel: lut[position of its InStamp's group in the stamp's list, exposure]
        lut = torch.as_tensor(np.stack([p[3] for p in per]).astype(np.int64), device=x.device)  # [B, 4, n_blk_expo]
        cs = torch.as_tensor(cumsum[:, 1:10].astype(np.int64), device=x.device)
        seg = torch.searchsorted(cs, torch.arange(keep, device=x.device).expand(len(chunk), keep).contiguous(), right=True).clamp_(max=8)
        lgp = torch.as_tensor(lg, device=x.device).gather(1, seg)  # group position of every pixel
        flat = lut.reshape(len(chunk), -1).gather(1, lgp * lut.shape[2] + expo[:, :keep].long())
        valid = torch.arange(keep, device=x.device)[None, :] < torch.as_tensor(n.astype(np.int64), device=x.device)[:, None]
        if bool(((flat < 0) & valid).any()):
            raise ValueError("a pixel belongs to an exposure its PSF group holds no PSF for (BlockTables.group_expo)")
        psf_slot = flat.clamp_(min=0).to(torch.int32)
    sb = StampBatch.from_device(cfg, tables, n, x[:, :keep], y[:, :keep], expo[:, :keep], indata[:, :, :keep],
                                [(i - 1) * cfg.n2 - cfg.fade for _, i in chunk], [(j - 1) * cfg.n2 - cfg.fade for j, _ in chunk],
                                n_expo, ctx=tables.ctx, psf_slot=psf_slot, maps=maps_)
    return sb



def coadd_block(cfg, pool, tables, n1P, n_expo, batch=64, pad_sides="", postage_pad=0, ldn=None, pipeline=True, stamps=None):
    """Coadd the n1P x n1P output stamps of a block.  ``pool``: InStampPool of the (n1P+2)^2 InStamps in row-major
    order (index j * nst + i, coadd.py:207); ``tables``: PSFGroupTables or BlockTables.  ``stamps``: the (j_st, i_st) to
    coadd (default all n1P x n1P, row by row as coadd.py:2049-2052); ``pad_sides=None`` leaves the boundary recovery of
    coadd.py:2163-2181 out.  Returns the BlockMaps."""
    nst = n1P + 2
    assert pool.n_inst == nst * nst
    maps = BlockMaps(n1P, cfg.n2, cfg.fade, cfg.n_inframe, n_expo, ctx=tables.ctx, device=str(pool.device),
                     n_out=int(getattr(tables, "n_out", 1)))
    todo = [(j, i) for j in range(1, n1P + 1) for i in range(1, n1P + 1)] if stamps is None else [(int(j), int(i)) for j, i in stamps]
    prepare = lambda chunk: prepare_batch(cfg, pool, tables, chunk, n1P, n_expo, ldn)  # noqa: E731

    # software pipeline: the next chunk is prepared on the host (and its selection / table kernels queued) right after
    # the current one's A and B builds have been queued, i.e. while the GPU is busy with them; only then does the host
    # block in the solve's status read-back
    chunks = [todo[c0 : c0 + batch] for c0 in range(0, len(todo), batch)]
    nxt = prepare(chunks[0]) if chunks else None
    for k, chunk in enumerate(chunks):
        sb = nxt
        sb.build()
        if pipeline:
            nxt = prepare(chunks[k + 1]) if k + 1 < len(chunks) else None
        sb.solve()
        sb.coadd()
        if not pipeline:
            nxt = prepare(chunks[k + 1]) if k + 1 < len(chunks) else None
        maps.add(sb.results(), [j for j, _ in chunk], [i for _, i in chunk])
    if pad_sides is not None:
        maps.finalize(pad_sides, postage_pad)
    return maps
