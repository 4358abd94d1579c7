"""Device-resident stamp path: PSF overlap tables -> A, B -> Cholesky solve -> maps -> coaddition.

Host-side mirror of the reference's stamp driver for a BATCH of independent postage stamps
(OutStamp._build_system_matrices + _perform_coaddition, reference src/pyimcom/coadd.py:1002-1122,
1294-1363; PSFGrp / PSFOvl, psfutil.py:520-1761).  torch is used only to own device memory and the
stream; every number is produced by libimcom_hip kernels through the C-ABI (include/imcom_hip.h).
"""

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np
import torch

from ._lib import PAIR_SWAP,  PAIR_FLIP, ImcomError, TableGeom, check, default_context, lib

NB = 128
SPECTRA_CHUNK = 1024  # PSFs sampled and transformed per call (BlockTables._ensure_spectra)


def _roundup(v, a):
    return (v + a - 1) // a * a


def _dp(t):
    return C.c_void_p(t.data_ptr())


def _hp(a):
    return a.ctypes.data_as(C.c_void_p)


def _runs(mask):
    """Runs [s0, s1) of consecutive true entries of a boolean mask."""
    m = np.concatenate([[False], np.asarray(mask, dtype=bool), [False]])
    d = np.flatnonzero(m[1:] != m[:-1])
    return [(int(a), int(b)) for a, b in zip(d[0::2], d[1::2])]


_GRIDS = {}


_UPLOAD = {}


def upload_stream(dev):
    """The stream host-to-device copies run on (one per device, highest priority), beside whatever the current stream is busy with."""
    dev = torch.device(dev)
    key = dev.index or 0
    if key not in _UPLOAD:
        _UPLOAD[key] = torch.cuda.Stream(dev, priority=int(os.environ.get("IMCOM_STREAM_PRIORITY", "-1")))
    return _UPLOAD[key]


def h2d(a, dev, dtype=None):
    """Host array -> device tensor through pinned memory without a host wait (a plain ``torch.as_tensor(a, device=dev)`` of pageable
    memory drains the stream first: 0.2-0.4 ms of idle GPU per call in a block).  The copy runs on a stream of its own and the
    current stream waits for it on the device: it does not queue behind the solve the current stream may be busy with, and its
    page-locked staging block is free again as soon as the copy is done (behind a long solve torch's host allocator found every
    block still in use and paid a fresh allocation, 2 ms, per call)."""
    dev = torch.device(dev)
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).pin_memory()
    up, cur = upload_stream(dev), torch.cuda.current_stream(dev)
    with torch.cuda.stream(up):
        out = t.to(dev, non_blocking=True)
    cur.wait_stream(up)
    out.record_stream(cur)
    return out


def free_device_bytes(dev):
    """Device memory a new allocation can draw on: what the driver reports free plus what torch's caching allocator holds
    without using it (mem_get_info alone shrinks with every buffer torch has cached, e.g. after an earlier block)."""
    free = torch.cuda.mem_get_info(dev)[0]
    return int(free + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev))


def _index_grid(na, nb):
    if (na, nb) not in _GRIDS:
        _GRIDS[(na, nb)] = tuple(np.ascontiguousarray(a) for a in np.meshgrid(np.arange(na), np.arange(nb), indexing="ij"))
    return _GRIDS[(na, nb)]


def psf_spectra(ctx, psf, nfft):
    """Forward spectra [n, size] (float64 view of complex [n, nfft/2+1, nfft]) of sampled PSFs [n, nsamp, nsamp] on the
    device, or None when nfft has no butterfly plan (imcom_psf_spectra_size == 0)."""
    n, ns, _ = psf.shape
    size = int(lib.imcom_psf_spectra_size(ns, nfft))
    if size == 0:
        return None
    spec = torch.empty((n, size), dtype=torch.float64, device=psf.device)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    check(lib.imcom_psf_spectra(ctx.handle, _dp(psf), n, ns, nfft, _dp(spec)))
    return spec


def overlap_tables(ctx, p1, s1, p2, s2, nsamp, nfft, pairs, amp, out, win=None, slots=None):
    """Tables of the (i, j) `pairs` between two PSF sets, from their spectra when both are given (else from the
    sampled PSFs through imcom_psf_overlap).  ``win`` [npairs, 4] (spectra form only): the part of every table's window
    that will be read, imcom_psf_overlap_spectra_win.  ``slots`` [npairs] (spectra form only): pair t goes to table
    ``out[slots[t]]`` of the arena ``out`` instead of ``out[t]``, imcom_psf_overlap_spectra_slots."""
    pairs = np.ascontiguousarray(pairs, dtype=np.int32)
    ampp = None if amp is None else _hp(amp)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    if s1 is not None and s2 is not None:
        winp = None
        if win is not None:
            win = np.ascontiguousarray(win, dtype=np.int32)
            assert win.shape == (len(pairs), 4)
            winp = _hp(win)
        slp = None
        if slots is not None:
            slots = np.ascontiguousarray(slots, dtype=np.int32)
            assert slots.shape == (len(pairs),)
            slp = _hp(slots)
        check(lib.imcom_psf_overlap_spectra_slots(ctx.handle, _dp(s1), s1.shape[0], _dp(s2), s2.shape[0], nsamp, nfft, _hp(pairs),
                                                  len(pairs), ampp, winp, slp, out.shape[0], _dp(out)))
    else:
        assert slots is None
        check(lib.imcom_psf_overlap(ctx.handle, _dp(p1), p1.shape[0], _dp(p2), p2.shape[0], nsamp, nfft, _hp(pairs), len(pairs),
                                    ampp, _dp(out)))


class PSFGroupTables:
    """Overlap tables of ONE input PSF group against itself and the target PSF(s).

    Mirrors PSFOvl(grp) (input self-overlap, triangle storage psfutil.py:1139-1175, 1270-1278),
    PSFOvl(grp, outgrp) (input-output, 1259-1265) and the output self-overlap value C (1283-1290).
    Table stack order: the E(E+1)/2 self tables in triangle order, then the E input-output tables of every target
    PSF (target-major).  ``psf_out`` [n_out, nsamp, nsamp]: the reference solves every target on its own
    (lakernel.py:121-128, kappa = kappaC * C of that target); ``Cs`` [n_out], ``C`` = Cs[0].
    """

    def __init__(self, psf_in, psf_out, nfft, ctx=None, device="cuda:0", amp_penalty=None):
        """amp_penalty: None or (cfg.amp_penalty[0], cfg.amp_penalty[1] * oversamp) (psfutil.py:661-671)."""
        self.ctx = ctx or default_context()
        amp = None if amp_penalty is None or 0.0 in tuple(amp_penalty) else np.array(amp_penalty, dtype=np.float64)
        E, ns, _ = psf_in.shape
        self.n_psf, self.nsamp, self.nfft = E, ns, nfft
        dev = torch.device(device)
        as_dev = lambda a: (a.to(dev).contiguous() if torch.is_tensor(a)  # noqa: E731
                            else torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev))
        pin, pout = as_dev(psf_in), as_dev(psf_out)
        O = self.n_out = pout.shape[0]
        ng = ns + 12
        self.ntri = E * (E + 1) // 2
        # one set of E + O PSFs: self pairs (triangle order), input-output pairs (target-major), output self pairs;
        # their spectra are computed once and every table comes from them in one call
        allp = torch.cat([pin, pout])
        spec = psf_spectra(self.ctx, allp, nfft)
        pairs = ([(i, j) for i in range(E) for j in range(i, E)] + [(i, E + o) for o in range(O) for i in range(E)]
                 + [(E + o, E + o) for o in range(O)])
        full = torch.empty((len(pairs), ng, ng), dtype=torch.float64, device=dev)
        overlap_tables(self.ctx, allp, spec, allp, spec, ns, nfft, pairs, amp, full)
        self.tables = full[: self.ntri + O * E]
        cc = full[self.ntri + O * E :]
        nc = ns // 2
        self.Cs = cc[:, 6 + nc, 6 + nc].cpu().numpy().astype(np.float64)  # psfutil.py:1290
        self.C = float(self.Cs[0])

    @classmethod
    def from_images(cls, psf_images, yxco, target, nsamp, nfft, oversamp, psf_circ=False, psf_norm=False, ctx=None,
                    device="cuda:0", amp_penalty=None):
        """The whole PSF side of a 2x2 stamp group on the device (PSFGrp.__init__ for the input group and for the
        output group, psfutil.py:615-671, then PSFOvl): ``psf_images`` [E, ny, nx] as returned by
        ``InImage.get_psf_pos``, ``yxco`` [E, 2, nsamp, nsamp] their sampling positions (psfutil.py:751-771, the
        WCS part stays on the host), ``target`` = (outpsf, extrasmooth, use_filter) of the configuration, or a list of
        such tuples (OUTPSF plus the cfg.outpsf_extra entries, psfutil.py:906-915)."""
        from . import psfs

        dev = torch.device(device)
        img = torch.as_tensor(np.ascontiguousarray(psf_images, dtype=np.float64), device=dev)
        co = torch.as_tensor(np.ascontiguousarray(yxco, dtype=np.float64), device=dev)
        psf_in = psfs.sample_psf(img, nsamp, co, psf_circ, psf_norm, ctx)
        targets = [target] if isinstance(target[0], str) else list(target)
        timg = torch.stack([psfs.get_outpsf(t[0], t[1], t[2], nsamp, oversamp, device=dev, ctx=ctx) for t in targets])
        psf_out = psfs.sample_psf(timg, nsamp, None, psf_circ, psf_norm, ctx)
        return cls(psf_in, psf_out, nfft, ctx=ctx, device=device, amp_penalty=amp_penalty)

    def _set_stream(self):
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    def tri(self, i, j):
        """PSFOvl._idx_square2triangle (psfutil.py:1175)."""
        assert i <= j
        return (2 * self.n_psf - i + 1) * i // 2 + j - i

    def pair_maps(self, flat_penalty):
        """pair_tab / pair_pen / io_tab for stamps whose pixels all belong to this group, local PSF
        index = exposure index (psfutil.py:1645-1708: table (j,i) for j <= i, the flipped (i,j) one
        otherwise; penalty -fp/n_psf, +fp on the same exposure)."""
        E = self.n_psf
        tab = np.zeros((E, E), np.int32)
        pen = np.zeros((E, E), np.float64)
        for a in range(E):
            for b in range(E):
                tab[a, b] = self.tri(a, b) if a <= b else (self.tri(b, a) | PAIR_FLIP)
                if flat_penalty != 0.0:
                    v = 0.0
                    v -= flat_penalty / E
                    if a == b:
                        v += flat_penalty
                    pen[a, b] = v
        io = np.arange(E, dtype=np.int32) + self.ntri
        return tab, pen, io

    def io_map(self, o):
        """io_tab of target PSF o (pair_maps returns the one of target 0)."""
        return np.arange(self.n_psf, dtype=np.int32) + self.ntri + o * self.n_psf


class BlockTables:
    """Overlap tables of a block whose PSFs vary from one 2x2 group of InStamps to the next (the reference's
    PSFGrp per InStamp pair of even indices, SysMatA.ji_st2psf psfutil.py:1803-1824).

    ``group_psfs``: {(gj, gi): array [n_g, nsamp, nsamp]} sampled input PSFs of every group; group (gj, gi) serves
    the InStamps (2gj..2gj+1, 2gi..2gi+1).  ``group_expo``: {(gj, gi): block exposure index of each of those PSFs}
    (PSFGrp.idx_grp2blk, psfutil.py:820-832: the reference keeps only exposures with pixels in the group; default
    0..n_g-1).  Table sets are computed on demand: a group's self overlap (triangle order) and input-output overlap, and
    the cross overlap of two groups (all n_g1 x n_g2 pairs, stored for the ordered key g_lo < g_hi).

    The tables live in ONE device arena of ``capacity`` tables (+ a zero table at index 0) that is handed out table by table
    (the builders address a table by its arena index, 64-bit element offsets: the arena may be as large as memory allows --
    a block of the reference's size, n1P = 84 with 6 exposures, works on ~12 k tables = 15 GB per batch of 252 stamps and
    would need 390 GB for all its sets at once).  ``capacity=None``: every set of the block if that fits into a third of the
    free device memory, else that third.  When the sets a batch needs do not fit next to the resident ones, ``on_full``
    decides: "evict" (default) frees the LEAST RECENTLY USED sets that the batch itself does not need -- what the reference's
    reference-counted sub-block cache does when an OutStamp is finished (psfutil.py:1868-1902) -- counted in ``evictions`` (calls
    that had to evict) and ``evicted_tables``; the kernels of earlier batches, queued on the same stream, have read their
    tables by the time the freed ones are rewritten.  "raise": ValueError instead.  The sets of ONE batch must fit, else
    ValueError (``blockrun.plan_batches`` sizes a block's batches by ``demand``)."""

    def __init__(self, group_psfs, psf_out, nfft, group_expo=None, capacity=None, amp_penalty=None, ctx=None, device="cuda:0", on_full="evict",
                 group_count=None, bulk_provider=None, cells=False, spec_capacity=None, eager_groups=False, provider_waits=False):
        assert on_full in ("evict", "raise")
        self.on_full, self.evictions, self.evicted_tables, self.computed_tables = on_full, 0, 0, 0
        self.ctx = ctx or default_context()
        dev = self.dev = torch.device(device)
        # sampled PSFs of a group: a device tensor (taken as it is), a host array (uploaded when the group is first needed: a
        # block's 81 groups are 570 MB at cfg-2 size), or -- with ``group_count`` = {key: number of PSFs} -- a callable that
        # returns the device tensor when the group is first needed (e.g. psfs.sample_psf on the resident PSF images: the
        # samples are then produced where they are consumed, PSFGrp.__init__ psfutil.py:640-656)
        # ``bulk_provider(keys)`` -> device tensor [sum of the keys' counts, nsamp, nsamp]: the sampled PSFs of SEVERAL groups in one
        # call (group_psfs may then map every key to None): a block's 81 groups are sampled and transformed by a handful of
        # launches instead of five small ones per group
        # ``cells=True``: the caller asserts that group (gj, gi) owns a cell of a grid of InStamps -- every pixel of a group with
        # smaller gi (gj) lies left of (below) every pixel of one with larger gi (gj), as for the reference's groups of 2 x 2
        # InStamps (coadd.py:207, 329-358).  The separations between two different groups' pixels then have one sign along
        # every axis in which the groups differ, and of their cross tables only that half (quarter) is computed.
        self.cells = bool(cells)
        # ``eager_groups=True``: the caller states that materialising a group costs the HOST nothing but queueing device work (PSF images
        # resident on the device, sampled there) -- coadd_block then asks for every group of a block at its start, so that the device
        # samples and transforms while the host plans the passes.  Not for providers that wait for host work (refblock's worker threads
        # deliver the groups in the order of the plan).
        self.eager_groups = bool(eager_groups)
        # ``provider_waits=True``: the opposite statement -- the provider may block on host work (refblock: PSF file broker and WCS on worker
        # threads).  coadd_block then asks ahead of time only for the groups of the FIRST pass; the next pass's groups are asked for when
        # that pass is prepared, i.e. while the device is busy with the current one.
        self.provider_waits = bool(provider_waits)
        self.psf = dict(group_psfs)
        self._order = {k: q for q, k in enumerate(self.psf)}
        self._bulk = bulk_provider
        self._count_of = {k: (int(group_count[k]) if (v is None or callable(v)) else int(v.shape[0])) for k, v in self.psf.items()}
        self.expo = {k: (list(range(self._count_of[k])) if group_expo is None else [int(e) for e in group_expo[k]]) for k in self.psf}
        assert all(len(self.expo[k]) == self._count_of[k] for k in self.psf)
        self.nsamp, self.nfft = int(psf_out.shape[-1]), nfft
        self.n_max = max(self._count_of.values())
        self.n_psf = self.n_max
        self.n_blk_expo = 1 + max(max(v) for v in self.expo.values())
        self.pout = (psf_out.to(dev, torch.float64).contiguous() if torch.is_tensor(psf_out)
                     else torch.as_tensor(np.ascontiguousarray(psf_out, dtype=np.float64), device=dev))
        O = self.n_out = self.pout.shape[0]
        amp = None if amp_penalty is None or 0.0 in tuple(amp_penalty) else np.array(amp_penalty, dtype=np.float64)
        self._amp = amp
        ng = self.nsamp + 12
        if capacity is None:
            third = int(free_device_bytes(dev) // 3 // (8 * ng * ng))
            capacity = max(min(self.block_demand(), third), 1)
        capacity = int(capacity)
        if capacity + 1 > (1 << 28):
            raise ValueError(f"capacity {capacity}: the pair codes of imcom_build_A carry 28-bit table indices")
        self.capacity = capacity
        # the arena; table 0 stays zero: the A builder lets samples without a stencil fetch the arena's first elements under
        # zero weights, and they must be finite whatever is (partly, with ``cells``) written elsewhere
        self.tables = torch.empty((capacity + 1, ng, ng), dtype=torch.float64, device=dev)
        self.tables[0].zero_()
        self.slots = {}                                                  # resident set -> arena indices of its tables (storage order)
        self._lru = {}                                                   # resident sets, least recently used first
        self._free = np.arange(capacity, 0, -1, dtype=np.int32)          # stack of free arena indices (handed out 1, 2, 3, ...)
        self._nfree = capacity
        # Forward spectra of the target PSFs and of the groups in use, in ONE arena of rows (a row = the spectrum of one PSF,
        # 4.7 MB at nfft 768), so that all the table sets a batch of stamps needs come out of a single call.  A group's rows are
        # assigned and filled when it is first needed; ``spec_capacity`` rows (default: the whole block if that fits into a sixth
        # of the free memory -- a block of the reference's size holds 1849 groups = 52 GB of spectra at 6 exposures): when they
        # are used up, every group the current request does not need is dropped and re-sampled / re-transformed if it comes
        # back (2 x 25 us per PSF).  size == 0: nfft has no butterfly plan, every set goes through imcom_psf_overlap on its own.
        size = int(lib.imcom_psf_spectra_size(self.nsamp, nfft))
        rows_all = O + sum(self._count_of.values())
        if spec_capacity is None and size:
            spec_capacity = max(min(rows_all, int(free_device_bytes(dev) // 6 // (8 * size))), O + 4 * self.n_max)
        self._spec_cap = min(rows_all, int(spec_capacity)) if size else 0
        self._spec_row, self._spec_next, self.spectra_resets = {None: 0}, O, 0
        self._spec_all = torch.empty((self._spec_cap, size), dtype=torch.float64, device=dev) if size else None
        self._dev_psf = {}  # device copies of the sampled PSFs, kept only when there are no spectra to keep instead
        if size:
            self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            check(lib.imcom_psf_spectra(self.ctx.handle, _dp(self.pout), O, self.nsamp, self.nfft, _dp(self._spec_all[:O])))
        cc = torch.empty((O, ng, ng), dtype=torch.float64, device=dev)
        self._compute([(None, None, [(o, o) for o in range(O)])], cc)
        nc = self.nsamp // 2
        self._Cs_dev, self._Cs = cc[:, 6 + nc, 6 + nc].contiguous(), None  # read back when first asked for: no host wait here

    @property
    def used(self):
        """Resident tables."""
        return self.capacity - self._nfree

    @property
    def Cs(self):
        if self._Cs is None:
            self._Cs = self._Cs_dev.cpu().numpy().astype(np.float64)
        return self._Cs

    @property
    def C(self):
        return float(self.Cs[0])

    def _sampled(self, gs):
        """Sampled PSFs of the groups `gs` on the device, [sum of their counts, nsamp, nsamp]."""
        todo = [g for g in gs if g not in self._dev_psf]
        got = {}
        if todo and self._bulk is not None and all(self.psf[g] is None for g in todo):
            p = self._bulk(todo)
            cnt = [self._count_of[g] for g in todo]
            assert tuple(p.shape) == (sum(cnt), self.nsamp, self.nsamp) and p.dtype == torch.float64 and p.is_contiguous(), \
                "bulk PSF provider returned the wrong shape"
            if len(todo) == len(gs):
                return p
            off = 0
            for g, c in zip(todo, cnt):
                got[g] = p[off : off + c]
                off += c
        for g in todo:
            if g in got:
                continue
            p = self.psf[g]
            if callable(p):
                p = p()
                assert tuple(p.shape) == (self._count_of[g], self.nsamp, self.nsamp), "group PSF provider returned the wrong shape"
            got[g] = (p.to(self.dev, torch.float64).contiguous() if torch.is_tensor(p)
                      else torch.as_tensor(np.ascontiguousarray(p, dtype=np.float64), device=self.dev))
        parts = [self._dev_psf[g] if g in self._dev_psf else got[g] for g in gs]
        return parts[0] if len(parts) == 1 else torch.cat(parts)

    def _psf_of(self, g):
        """Device tensor of a group's sampled PSFs, cached (the path without spectra)."""
        if g is None:
            return self.pout
        if g not in self._dev_psf:
            self._dev_psf[g] = self._sampled([g])
        return self._dev_psf[g]

    def _ensure_spectra(self, gs):
        """Sample / upload and transform the groups of `gs` whose spectra are not resident, in one call."""
        gs = [g for g in dict.fromkeys(gs) if g is not None]
        miss = [g for g in gs if g not in self._spec_row]
        if not miss:
            return
        if self._spec_next + sum(self._count_of[g] for g in miss) > self._spec_cap:
            # out of rows: drop every group this request does not need (earlier requests' table kernels are queued ahead of the
            # refill on the same stream)
            self._spec_row, self._spec_next = {None: 0}, self.n_out
            self.spectra_resets += 1
            miss = gs
            if self._spec_next + sum(self._count_of[g] for g in miss) > self._spec_cap:
                raise ValueError(f"spectra arena of {self._spec_cap} rows cannot hold the PSF groups of one batch")
        miss.sort(key=self._order.__getitem__)
        # in runs of at most SPECTRA_CHUNK PSFs: the sampled images are a transient of 1.2 MB per PSF and the transform takes 2.4 MB of
        # workspace per PSF (a block's 11 k PSFs in one call would be 13 + 26 GB that no plan accounts for)
        q0 = 0
        while q0 < len(miss):
            q1, cnt = q0, 0
            while q1 < len(miss) and (q1 == q0 or cnt + self._count_of[miss[q1]] <= SPECTRA_CHUNK):
                cnt += self._count_of[miss[q1]]
                q1 += 1
            part = miss[q0:q1]
            p = self._sampled(part)
            r0 = self._spec_next
            self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            check(lib.imcom_psf_spectra(self.ctx.handle, _dp(p), p.shape[0], self.nsamp, self.nfft, _dp(self._spec_all[r0 : r0 + p.shape[0]])))
            for g in part:
                self._spec_row[g] = self._spec_next
                self._spec_next += self._count_of[g]
            q0 = q1

    def prefetch(self, groups):
        """Queue the sampling / spectra of `groups` now (as far as the spectra arena holds them; otherwise groups are
        materialised when first needed)."""
        if self._spec_all is None:
            return
        gs, rows = [], self._spec_next
        for g in sorted((g for g in groups if g in self.psf and g not in self._spec_row), key=self._order.__getitem__):
            rows += self._count_of[g]
            if rows > self._spec_cap:
                break
            gs.append(g)
        self._ensure_spectra(gs)

    def _compute(self, jobs, out, slots=None):
        """jobs: list of (g1, g2, local pairs) (g = None: the target PSFs).  Their tables fill `out` back to back, or -- with
        ``slots`` (int32, one arena index per table) -- the arena tables ``out[slots]``."""
        if self._spec_all is None:
            off = 0
            for g1, g2, pairs in jobs:
                dst = out[off : off + len(pairs)] if slots is None else torch.empty((len(pairs),) + tuple(out.shape[1:]), dtype=out.dtype, device=out.device)
                overlap_tables(self.ctx, self._psf_of(g1), None, self._psf_of(g2), None, self.nsamp, self.nfft, pairs, self._amp, dst)
                if slots is not None:
                    out.index_copy_(0, h2d(slots[off : off + len(pairs)], out.device, np.int64), dst)
                off += len(pairs)
            return
        allp, allw = [], []
        ns, nc, margin = self.nsamp, self.nsamp // 2, 8  # ten-tap stencils reach 4 below / 5 above the cell of a separation
        rng = {-1: (0, min(ns, nc + margin)), 0: (0, ns), 1: (max(0, nc - margin), ns)}
        sgn = lambda a, b: (a > b) - (a < b)  # noqa: E731
        self._ensure_spectra([g for g1, g2, _ in jobs for g in (g1, g2)])
        for g1, g2, pairs in jobs:
            allp.append(np.asarray(pairs, dtype=np.int64).reshape(-1, 2) + (self._spec_row[g1], self._spec_row[g2]))
            if self.cells:
                # tables are read at r(g1) - r(g2) (A builder: the group that sorts first on the left; input-output and self: any sign)
                sy, sx = (0, 0) if g1 is None or g2 is None or g1 == g2 else (sgn(g1[0], g2[0]), sgn(g1[1], g2[1]))
                allw.append(np.broadcast_to(np.array(rng[sy] + rng[sx], dtype=np.int32), (len(pairs), 4)))
        overlap_tables(self.ctx, None, self._spec_all, None, self._spec_all, self.nsamp, self.nfft, np.concatenate(allp), self._amp, out,
                       win=np.concatenate(allw) if self.cells else None, slots=slots)

    def _n(self, g):
        return self._count_of[g]

    _pair_memo = {}

    @classmethod
    def _local_pairs(cls, kind, n1, n2):
        """(i, j) PSF pairs of a table set in storage order, [count, 2] int64 (shared between all sets of the same shape)."""
        key = (kind, n1, n2)
        if key not in cls._pair_memo:
            if kind == "self":
                p = [(i, j) for i in range(n1) for j in range(i, n1)]
            elif kind == "io":
                p = [(i, o) for o in range(n2) for i in range(n1)]
            else:
                p = [(i, j) for i in range(n1) for j in range(n2)]
            cls._pair_memo[key] = np.asarray(p, dtype=np.int64).reshape(-1, 2)
        return cls._pair_memo[key]

    def _count(self, key):
        if key[0] == "self":
            return self._n(key[1]) * (self._n(key[1]) + 1) // 2
        if key[0] == "io":
            return self.n_out * self._n(key[1])
        return self._n(key[1]) * self._n(key[2])

    def demand(self, keys):
        """Tables the sets `keys` take together (what one batch of stamps needs resident at once)."""
        return sum(self._count(k) for k in dict.fromkeys(keys))

    def block_demand(self):
        """Tables of every set of the block: self and input-output of every group, cross of every pair of neighbouring groups."""
        gs = set(self.psf)
        keys = [("self", g) for g in gs] + [("io", g) for g in gs]
        keys += [("cross", g, h) for g in gs for h in ((g[0], g[1] + 1), (g[0] + 1, g[1] - 1), (g[0] + 1, g[1]), (g[0] + 1, g[1] + 1)) if h in gs]
        return self.demand(keys)

    def _release(self, key):
        sl = self.slots.pop(key)
        del self._lru[key]
        self._free[self._nfree : self._nfree + len(sl)] = sl[::-1]
        self._nfree += len(sl)
        return len(sl)

    def drop_all(self):
        """Forget every resident set (the arena keeps its memory)."""
        for k in list(self.slots):
            self._release(k)

    def reset(self):
        """Forget every table set, every group's spectra and cached samples: the next block (or the next repetition of a
        timing loop) computes everything again in the memory this object already holds."""
        self.drop_all()
        self._free[:] = np.arange(self.capacity, 0, -1, dtype=np.int32)
        self._spec_row, self._spec_next = {None: 0}, self.n_out
        self._dev_psf = {}
        self.evictions = self.evicted_tables = self.computed_tables = self.spectra_resets = 0

    def require(self, keys):
        """Make sure the table sets `keys` are in the arena; returns {key: arena indices of its tables, storage order}."""
        keys = list(dict.fromkeys(keys))
        missing = [k for k in keys if k not in self.slots]
        need = sum(self._count(k) for k in missing)
        for k in keys:
            if k in self._lru:  # most recently used last
                self._lru[k] = self._lru.pop(k)
        if need > self._nfree:
            total = self.demand(keys)
            if total > self.capacity:
                raise ValueError(f"table arena of {self.capacity} tables cannot hold the {total} of one batch")
            if self.on_full == "raise":
                raise ValueError(f"table arena full: {self.used} of {self.capacity} tables resident, {need} more needed (on_full='raise')")
            wanted = set(keys)
            for k in [k for k in self._lru if k not in wanted]:  # least recently used first
                self.evicted_tables += self._release(k)
                if need <= self._nfree:
                    break
            self.evictions += 1
        if missing:
            jobs = []
            first = self._nfree - need
            sl_all = self._free[first : self._nfree][::-1].copy()  # lowest stack entries last: indices come out in hand-out order
            self._nfree = first
            off = 0
            for k in missing:
                if k[0] == "self":
                    jobs.append((k[1], k[1], self._local_pairs("self", self._n(k[1]), 0)))
                elif k[0] == "io":
                    jobs.append((k[1], None, self._local_pairs("io", self._n(k[1]), self.n_out)))
                else:
                    jobs.append((k[1], k[2], self._local_pairs("cross", self._n(k[1]), self._n(k[2]))))
                c = self._count(k)
                self.slots[k] = sl_all[off : off + c]
                self._lru[k] = None
                off += c
            self._compute(jobs, self.tables, slots=sl_all)
            self.computed_tables += need
        return {k: self.slots[k] for k in keys}

    def tables_of(self, key):
        """The tables of a resident set as one tensor [count, ng, ng] (a copy, gathered from the arena)."""
        return self.tables.index_select(0, h2d(self.slots[key], self.dev, np.int64))

    @staticmethod
    def keys_for(groups):
        """Table sets a stamp touching `groups` (distinct group keys) needs."""
        gs = sorted(set(groups))
        return ([("self", g) for g in gs] + [("io", g) for g in gs] + [("cross", a, b) for i, a in enumerate(gs) for b in gs[i + 1 :]])

    def stamp_maps(self, groups, flat_penalty, slots=4):
        """Maps of one stamp whose pixels belong to `groups` (list of <= `slots` distinct group keys): pair_tab
        [P, P], pair_pen [P, P], io_tab [P] ([n_out, P] with several target PSFs) with P = slots * n_max, and lut [slots, n_blk_expo] = stamp-local PSF
        index of a pixel of (position of its group in the list, block exposure), -1 where the group lacks the
        exposure.  Call require() first."""
        P = slots * self.n_max
        tab = np.full((P, P), -1, np.int32)
        pen = np.zeros((P, P))
        io = np.zeros((self.n_out, P), np.int32)
        lut = np.full((slots, self.n_blk_expo), -1, np.int32)
        base = np.concatenate([[0], np.cumsum([self._n(g) for g in groups])]).astype(int)
        for la, ga in enumerate(groups):
            na = self._n(ga)
            ea = np.asarray(self.expo[ga])
            lut[la, ea] = base[la] + np.arange(na)
            io[:, base[la] : base[la] + na] = self.slots[("io", ga)].reshape(self.n_out, na)
            for lb, gb in enumerate(groups):
                nb = self._n(gb)
                ka, kb = _index_grid(na, nb)
                if ga == gb:  # triangle storage of the group's self overlap (psfutil.py:1175); (j, i) with j > i flipped
                    lo, hi = np.minimum(ka, kb), np.maximum(ka, kb)
                    blk = self.slots[("self", ga)][(2 * na - lo + 1) * lo // 2 + hi - lo]
                    blk = np.where(ka <= kb, blk, blk | PAIR_FLIP)
                elif ga < gb:
                    blk = self.slots[("cross", ga, gb)][ka * nb + kb]
                else:  # evaluated from the other group's side (psfutil.py:1990-1996)
                    blk = self.slots[("cross", gb, ga)][kb * na + ka] | PAIR_SWAP
                tab[base[la] : base[la] + na, base[lb] : base[lb] + nb] = blk
                if flat_penalty != 0.0:  # psfutil.py:1433, 1482-1486, 1705-1708
                    same = ea[:, None] == np.asarray(self.expo[gb])[None, :]
                    pen[base[la] : base[la] + na, base[lb] : base[lb] + nb] = -flat_penalty / (na * nb) ** 0.5 + np.where(same, flat_penalty, 0.0)
        return tab, pen, (io[0] if self.n_out == 1 else io), lut


class BatchBuffers:
    """Device memory for the large per-batch arrays (A, -B/2, T: 27 GB for 256 cfg-2 stamps), kept from one batch of a
    block to the next: a block's batches differ in their leading dimension, so torch's caching allocator would satisfy none of
    them from the previous batch's blocks (measured: 100 ms of hipMalloc per batch).  ``take`` returns a view of the named
    flat buffer, which grows when a batch needs more than any before it."""

    def __init__(self, device):
        self.dev, self._flat = torch.device(device), {}

    def nbytes(self):
        """Bytes this set holds (a later batch reuses them: a planner counts them as available)."""
        return sum(f.numel() * f.element_size() for f in self._flat.values() if f is not None)

    def take(self, name, shape, dtype):
        n = int(np.prod(shape))
        f = self._flat.get(name)
        if f is None or f.dtype != dtype or f.numel() < n:
            self._flat[name] = None  # drop the old block before the larger one is requested
            if f is not None:
                # ... and hand it back to the driver: torch's allocator would keep it cached, of no use to a larger request and out of
                # reach for the library's own workspace allocation (a pass that then does not fit is halved, blockrun.coadd_block)
                torch.cuda.empty_cache()
            f = self._flat[name] = torch.empty(n, dtype=dtype, device=self.dev)
        return f[:n].view(*shape)


@dataclass
class StampBatchResult:
    """Outputs of a batch for ONE target PSF (StampBatch.results() lists them for n_out > 1)."""
    Tt: torch.Tensor          # [batch, ldn, ldm] float32, input-pixel-major T (tapered when fade > 0)
    UC: torch.Tensor          # [batch, n2f, n2f] float32
    Sigma: torch.Tensor
    kappa: torch.Tensor
    outimage: torch.Tensor    # [batch, n_inframe, n2f, n2f] float32
    Tsum_stamp: torch.Tensor  # [batch, n_expo] float64
    Tsum_inpix: torch.Tensor  # [batch, n2f, n2f] float64
    Neff: torch.Tensor
    info: np.ndarray
    n: np.ndarray
    perm: torch.Tensor = None  # [batch, ldn] or None: the batch's pixel order (blockrun.prepare_batch orders a stamp's pixels by PSF)

    def T(self, s, order="reference"):
        """T of stamp s in the reference layout [m, N] (lakernel.py:96), input pixels in the reference's order (coadd.py:937);
        ``order="batch"``: in the order the batch holds them (the order of its x, y, A, -B/2)."""
        m = self.UC.shape[-1] * self.UC.shape[-2]
        n = int(self.n[s])
        Tm = self.Tt[s, :n, :m].T.contiguous()
        if self.perm is None or order == "batch":
            return Tm
        out = torch.empty_like(Tm)
        out[:, self.perm[s, :n]] = Tm
        return out

    def to_reference_order(self, s, a, axes=(0,)):
        """An array of stamp s whose `axes` run over the batch's input pixels (rows / columns of A, rows of -B/2 ...), in the reference's order."""
        n = int(self.n[s])
        if self.perm is None:
            return a
        inv = torch.empty(n, dtype=torch.long, device=self.perm.device)
        inv[self.perm[s, :n]] = torch.arange(n, device=self.perm.device)
        for ax in axes:
            a = a.index_select(ax, inv)
        return a


class StampBatch:
    """A batch of stamps that share one PSF group, resident on one GPU.

    Upload once (constructor), then ``build()`` (A and B), ``solve()``, ``coadd()`` or ``run()`` for all
    three; buffers are reused across calls so a timed loop allocates nothing.  With n_out > 1 target PSFs
    (``tables.n_out``) A is built once and B, the solve (kappa = kappaC * C of the target, lakernel.py:121-128) and
    the coaddition run per target; every per-target buffer ``X_o`` has a leading target axis and ``X`` is target 0.
    """

    def __init__(self, cfg, stamps, tables: PSFGroupTables, ctx=None, device="cuda:0", ldn=None):
        dev = torch.device(device)
        B = len(stamps)
        n = np.array([s.n for s in stamps], dtype=np.int32)
        ldn = ldn or _roundup(max(int(n.max()), 1), NB)
        x = np.zeros((B, ldn)); y = np.zeros((B, ldn))
        psf = np.zeros((B, ldn), np.int32)
        indata = np.zeros((B, cfg.n_inframe, ldn), np.float32)
        for b, s in enumerate(stamps):
            x[b, : s.n], y[b, : s.n], psf[b, : s.n] = s.x, s.y, s.expo
            indata[b, :, : s.n] = s.indata
        self._setup(cfg, tables, ctx, dev, n, ldn, max(s.n_expo for s in stamps), torch.as_tensor(x, device=dev),
                    torch.as_tensor(y, device=dev), torch.as_tensor(psf, device=dev), torch.as_tensor(indata, device=dev),
                    np.array([s.out_x0 for s in stamps], dtype=np.float64), np.array([s.out_y0 for s in stamps], dtype=np.float64))

    @classmethod
    def from_device(cls, cfg, tables, n, x, y, expo, indata, out_x0, out_y0, n_expo, ctx=None, psf_slot=None, maps=None, buffers=None):
        """Batch whose pixel lists are already on the GPU (e.g. from pyimcom_amd.select.select_pixels): x, y f64 and
        expo i32 [B, ldn], indata f32 [B, n_inframe, ldn] with ldn a multiple of 128 and zero padding; n, out_x0,
        out_y0 host arrays [B].  Stamps whose pixels belong to several PSF groups pass ``psf_slot`` (i32 [B, ldn],
        stamp-local PSF index of each pixel) and ``maps`` = (pair_tab [B,P,P] i32, pair_pen [B,P,P] f64, io_tab [B,P] i32)
        as documented at imcom_build_A / imcom_build_B; ``tables`` then only needs .tables, .nsamp, .C, .ctx.  ``buffers``: a
        BatchBuffers the large arrays are taken from (they are then valid until its next use)."""
        self = cls.__new__(cls)
        ldn = x.shape[1]
        assert ldn % NB == 0 and tuple(indata.shape) == (x.shape[0], cfg.n_inframe, ldn)
        self._setup(cfg, tables, ctx, x.device, np.ascontiguousarray(n, dtype=np.int32), ldn, int(n_expo), x.contiguous(), y.contiguous(),
                    expo.contiguous(), indata.contiguous(), np.ascontiguousarray(out_x0, dtype=np.float64),
                    np.ascontiguousarray(out_y0, dtype=np.float64), psf_slot=psf_slot, maps=maps, buffers=buffers)
        return self

    def _setup(self, cfg, tables, ctx, dev, n, ldn, n_expo, x, y, expo, indata, out_x0, out_y0, psf_slot=None, maps=None, buffers=None):
        self.cfg, self.tables = cfg, tables
        self.ctx = ctx or tables.ctx
        self.dev = dev
        self.batch = B = len(n)
        self.n, self.ldn = n, ldn
        self.m, self.n2f = cfg.m, cfg.n2f
        self.ldm = _roundup(self.m, NB)
        self.n_expo = n_expo
        f64, f32 = torch.float64, torch.float32
        # expo: exposure (input image) of each pixel, what the coaddition sums over; psf: stamp-local PSF index, what
        # selects the overlap tables.  One PSF group: the two coincide.
        self.x, self.y, self.expo, self.indata = x, y, expo, indata
        self.psf = expo if psf_slot is None else psf_slot.contiguous()
        self.out_x0 = h2d(out_x0, dev)
        self.out_y0 = h2d(out_y0, dev)
        O = self.n_out = int(getattr(tables, "n_out", 1))
        if maps is None:
            assert self.n_expo <= tables.n_psf
            tab, pen, _ = tables.pair_maps(cfg.flat_penalty)
            P = self.npsf = tables.n_psf
            self.pair_tab = torch.as_tensor(np.broadcast_to(tab, (B, P, P)).copy(), device=dev)
            self.pair_pen = torch.as_tensor(np.broadcast_to(pen, (B, P, P)).copy(), device=dev)
            io = np.stack([np.broadcast_to(tables.io_map(o), (B, P)) for o in range(O)])
        else:
            tab, pen, io = maps
            P = self.npsf = tab.shape[-1]
            io = np.asarray(io)
            if io.ndim == 3 and io.shape[0] == B and O > 1:  # [B, n_out, P] as stacked from BlockTables.stamp_maps
                io = io.transpose(1, 0, 2)
            io = io.reshape(O, B, P)
            assert tab.shape == (B, P, P) and pen.shape == (B, P, P)
            self.pair_tab = h2d(tab, dev, np.int32)
            self.pair_pen = h2d(pen, dev, np.float64)
        self.io_tab_o = h2d(io, dev, np.int32)  # [n_out, B, P]
        self.geom = TableGeom(tables.nsamp, float(tables.nsamp // 2), float(cfg.dscale), float(cfg.flat_penalty))
        Cs = np.asarray(getattr(tables, "Cs", [tables.C]), dtype=np.float64)
        self.Cs_o = np.ascontiguousarray(np.broadcast_to(Cs[:, None], (O, B)))
        self.kappaC = np.ascontiguousarray(cfg.kappaC, dtype=np.float64)
        # device buffers
        big = buffers.take if buffers is not None else (lambda name, shape, dtype: torch.empty(shape, dtype=dtype, device=dev))
        self.A = big("A", (B, self.ldn, self.ldn), f64)
        self.Bt_o = big("Bt", (O, B, self.ldn, self.ldm), f64)
        self.Tt_o = big("Tt", (O, B, self.ldn, self.ldm), f32)
        self.UC_o = torch.empty((O, B, self.m), dtype=f32, device=dev)
        self.Sigma_o = torch.empty((O, B, self.m), dtype=f32, device=dev)
        self.kappa_o = torch.empty((O, B, self.m), dtype=f32, device=dev)
        self.outimage_o = torch.empty((O, B, cfg.n_inframe, self.m), dtype=f32, device=dev)
        self.Tsum_stamp_o = torch.empty((O, B, self.n_expo), dtype=f64, device=dev)
        self.Tsum_inpix_o = torch.empty((O, B, self.m), dtype=f64, device=dev)
        self.Neff_o = torch.empty((O, B, self.m), dtype=f64, device=dev)
        self.info_o = np.zeros((O, B), np.int32)
        # target 0 under the plain names
        self.io_tab, self.Cs, self.info = self.io_tab_o[0], self.Cs_o[0], self.info_o[0]
        for name in ("Bt", "Tt", "UC", "Sigma", "kappa", "outimage", "Tsum_stamp", "Tsum_inpix", "Neff"):
            setattr(self, name, getattr(self, name + "_o")[0])

    def _stream(self):
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    def build(self):
        """A (psfutil.py:1401-1495, 1597-1732 + coadd.py:1027-1068) and B (1497-1595 + coadd.py:1075-1082)."""
        if self._no_qlt_ctrl():
            return  # coadd.py:1020-1025: the Empirical kernel without quality control never builds the system matrices
        self._stream()
        h = self.ctx.handle
        check(lib.imcom_build_A(h, self.batch, _hp(self.n), self.ldn, _dp(self.x), _dp(self.y), _dp(self.psf),
                                _dp(self.tables.tables), self.tables.tables.shape[0], C.byref(self.geom),
                                _dp(self.pair_tab), _dp(self.pair_pen), self.npsf, _dp(self.A)))
        for o in range(self.n_out):
            check(lib.imcom_build_B(h, self.batch, _hp(self.n), self.ldn, _dp(self.x), _dp(self.y), _dp(self.psf),
                                    _dp(self.tables.tables), self.tables.tables.shape[0], C.byref(self.geom),
                                    _dp(self.io_tab_o[o]), self.npsf, _dp(self.out_x0), _dp(self.out_y0), self.n2f, self.ldm,
                                    _dp(self.Bt_o[o])))

    def _no_qlt_ctrl(self):
        return self.cfg.kernel == "Empirical" and bool(getattr(self.cfg, "no_qlt_ctrl", False))

    def solve(self):
        """lakernel.CholKernel (lakernel.py:281-394) + the map taper of coadd.py:1118-1122.  With IMCOM_EPILOGUE_FUSED=1 (Cholesky,
        fade 0) the coaddition of the same stamps rides in the call (imcom_solve_chol_resident_coadd: with one kappa node its sums
        are taken from the tiles of T inside the backward launches) and ``coadd()`` finds it done -- measured: the 2.0 ms the pass
        over T costs per 256 cfg-2 stamps come back as 2.0 ms more in the backward launches, so it is not the default."""
        self._stream()
        self._coadded = set()
        for o in range(self.n_out):
            try:
                self._solve_target(self.Bt_o[o], self.Cs_o[o], self.Tt_o[o], self.UC_o[o], self.Sigma_o[o], self.kappa_o[o], self.info_o[o], o)
            except ImcomError as e:
                if e.status != -3:  # IMCOM_ERR_NOMEM: the library's workspace is a device allocation of its own -- memory that torch's
                    raise           # caching allocator holds without using it is not available to it until it is handed back
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
                self._solve_target(self.Bt_o[o], self.Cs_o[o], self.Tt_o[o], self.UC_o[o], self.Sigma_o[o], self.kappa_o[o], self.info_o[o], o)

    def _repair_hint(self, hint):
        """Hand the driver's estimate of max |w[0]| to the context for the calls of this solve (None / 0: none)."""
        self.ctx.set_repair_hint(hint if hint and np.isfinite(hint) and hint > 0 else 0.0)

    def _note_repair(self):
        """What the solve's repaired stamps had: ``repair_absmax`` = max |w[0]| (None when nothing was repaired), the next pass's hint."""
        cnt, lo, hi = self.ctx.last_repair()
        self.repair_absmax = max(abs(lo), abs(hi)) if cnt else None

    def solve_begin(self, expect_repair=False, repair_hint=None):
        """The solve queued, not waited for (imcom_solve_chol_resident_begin; one target PSF, Cholesky kernel -- for anything else nothing
        happens here and ``solve_end()`` runs the synchronous ``solve()``): the caller prepares its next pass while the device factors and
        solves, then calls ``solve_end()``.  ``expect_repair``: the caller has seen (nearly) every stamp of the previous pass take
        _cholesky_wrapper's repair (lakernel.py:262-279; the reference's production shape: DESIGN.md section 4): the factorisation that
        would fail is not attempted -- ``solve_end()`` goes straight to the smallest eigenvalues, which also say for every stamp whether
        A + kappa I is positive definite after all (such a stamp is then factored plainly, as the reference would have).
        ``repair_hint``: max |w[0]| of the stamps the previous pass repaired (``repair_absmax`` of that pass): the smallest-eigenvalue
        iteration of this pass's repairs starts there (imcom_ctx_set_repair_hint) -- its path changes, not what it converges to."""
        cfg = self.cfg
        self._deferred, self._unsolved, self._expected = False, False, False
        self.repair_absmax = None
        self._repair_hint(repair_hint)
        if expect_repair and cfg.kernel == "Cholesky" and self.n_out == 1 and len(self.kappaC) == 1 and os.environ.get("IMCOM_REDO_ALL") != "1":
            self._expected = True
            return
        if cfg.kernel != "Cholesky" or self.n_out != 1 or (cfg.fade == 0 and os.environ.get("IMCOM_EPILOGUE_FUSED") == "1") \
                or os.environ.get("IMCOM_SOLVE_DEFERRED", "1") == "0":
            self._unsolved = True  # solve_end() runs the synchronous solve: the caller's work in between still overlaps the builds
            return
        self._stream()
        self._coadded = set()
        rc = lib.imcom_solve_chol_resident_begin(self.ctx.handle, self.batch, _hp(self.n), self.ldn, self.m, self.ldm, _dp(self.A), _dp(self.Bt_o[0]),
                                                 _hp(self.Cs_o[0]), _hp(self.kappaC), len(self.kappaC), float(cfg.uctarget), float(cfg.sigmamax),
                                                 _dp(self.Tt_o[0]), _dp(self.UC_o[0]), _dp(self.Sigma_o[0]), _dp(self.kappa_o[0]))
        if rc == -3:  # the workspace could not grow (nothing has been queued): the synchronous path hands torch's cached memory back and retries
            self._unsolved = True
            return
        check(rc)
        if cfg.fade > 0:
            for t in (self.kappa_o[0], self.Sigma_o[0], self.UC_o[0]):
                check(lib.imcom_trapezoid_f32(self.ctx.handle, _dp(t), self.batch, self.n2f, cfg.fade))
        self._deferred = True

    def solve_end(self):
        """Wait for ``solve_begin()``'s work.  Stamps whose factorisation failed are solved again with the reference's repair
        (lakernel.py:262-279) -- those stamps only (imcom_solve_chol_resident_redo; the others' outputs are final).  Returns False, or
        the boolean mask [batch] of the stamps that were solved again: whatever the caller queued on the first attempt's outputs of
        THOSE stamps (the coaddition) has to be queued again for them (``coadd(only=mask)``).  ``repair_share`` and ``repair_absmax``
        then say what the next pass may expect."""
        try:
            return self._solve_end()
        finally:
            if self.ctx._h:
                self.ctx.set_repair_hint(0.0)  # (the hint was for this solve: a later call on the context starts without one)

    def _solve_end(self):
        if getattr(self, "_unsolved", False):
            self._unsolved = False
            self.solve()
            self._note_repair()
            return False
        if getattr(self, "_expected", False):
            self._expected = False
            self._stream()
            self._coadded = set()
            cfg, info = self.cfg, self.info_o[0]
            redo = np.full(self.batch, 3, dtype=np.int32)  # 3: the failure is this driver's EXPECTATION (2: observed by solve_end, below)
            check(lib.imcom_solve_chol_resident_redo(self.ctx.handle, self.batch, _hp(self.n), self.ldn, self.m, self.ldm, _dp(self.A), _dp(self.Bt_o[0]),
                                                     _hp(self.Cs_o[0]), _hp(self.kappaC), 1, float(cfg.uctarget), float(cfg.sigmamax),
                                                     _dp(self.Tt_o[0]), _dp(self.UC_o[0]), _dp(self.Sigma_o[0]), _dp(self.kappa_o[0]), _hp(redo), _hp(info)))
            if cfg.fade > 0:
                for t in (self.kappa_o[0], self.Sigma_o[0], self.UC_o[0]):
                    check(lib.imcom_trapezoid_f32(self.ctx.handle, _dp(t), self.batch, self.n2f, cfg.fade))
            self.repair_share = float((info != 0).mean())
            self._note_repair()
            return False
        if not getattr(self, "_deferred", False):
            return False
        self._deferred = False
        self._stream()
        cfg, info = self.cfg, self.info_o[0]
        rc = lib.imcom_solve_chol_resident_end(self.ctx.handle, self.batch, _hp(info))
        if rc == 1:
            again = info != 0
            if len(self.kappaC) != 1 or os.environ.get("IMCOM_REDO_ALL") == "1":  # several kappa nodes: every stamp again (the synchronous entry)
                self.solve()
                return np.ones(self.batch, dtype=bool)
            redo = np.where(again, 2, 0).astype(np.int32)
            check(lib.imcom_solve_chol_resident_redo(self.ctx.handle, self.batch, _hp(self.n), self.ldn, self.m, self.ldm, _dp(self.A), _dp(self.Bt_o[0]),
                                                     _hp(self.Cs_o[0]), _hp(self.kappaC), 1, float(cfg.uctarget), float(cfg.sigmamax),
                                                     _dp(self.Tt_o[0]), _dp(self.UC_o[0]), _dp(self.Sigma_o[0]), _dp(self.kappa_o[0]), _hp(redo), _hp(info)))
            if cfg.fade > 0:  # the map tapers of coadd.py:1118-1122 on the stamps that were solved again (the others have theirs)
                for s0, s1 in _runs(again):
                    for t in (self.kappa_o[0], self.Sigma_o[0], self.UC_o[0]):
                        check(lib.imcom_trapezoid_f32(self.ctx.handle, _dp(t[s0:]), s1 - s0, self.n2f, cfg.fade))
            self.repair_share = float(again.mean())
            self._note_repair()
            return again
        check(rc)
        self.repair_share = 0.0
        return False

    def _solve_target(self, Bt, Cs, Tt, UC, Sigma, kappa, info, o=0):
        cfg = self.cfg
        if cfg.kernel == "Eigen":
            # lakernel.EigenKernel (lakernel.py:141-223) on the resident layouts
            check(lib.imcom_solve_eigen_resident(self.ctx.handle, self.batch, _hp(self.n), self.ldn, self.m, self.ldm, _dp(self.A), _dp(Bt),
                                                 _hp(Cs), _hp(self.kappaC), len(self.kappaC), float(cfg.uctarget), float(cfg.sigmamax), 13,
                                                 _dp(Tt), _dp(UC), _dp(Sigma), _dp(kappa), _hp(info)))
        elif cfg.kernel in ("Iterative", "Empirical"):
            # lakernel.IterKernel / EmpirKernel (lakernel.py:533-805) on device pointers; the output pixel centres
            # are the integer grid starting at (out_y0, out_x0), the acceptance radius is INPAD in output pixels
            nqc = self._no_qlt_ctrl()
            mB = None if nqc else Bt[:, :, : self.m].transpose(1, 2).contiguous()
            T = torch.empty((self.batch, self.m, self.ldn), dtype=torch.float32, device=self.dev)
            g = torch.arange(self.n2f, dtype=torch.float64, device=self.dev)
            oy = (self.out_y0[:, None, None] + g[None, :, None]).expand(self.batch, self.n2f, self.n2f)
            ox = (self.out_x0[:, None, None] + g[None, None, :]).expand(self.batch, self.n2f, self.n2f)
            yx = torch.stack([oy.reshape(self.batch, self.m), ox.reshape(self.batch, self.m)], dim=1).contiguous()
            if cfg.kernel == "Iterative":
                nv = len(self.kappaC)
                check(lib.imcom_solve_iter(self.ctx.handle, self.batch, _hp(self.n), self.ldn, self.m, _dp(self.A), _dp(mB),
                                           _hp(Cs), _hp(self.kappaC), nv, float(cfg.uctarget), float(cfg.sigmamax), _dp(yx),
                                           _dp(self.y), _dp(self.x), float(cfg.rho), float(getattr(cfg, "iter_rtol", 1.5e-3)),
                                           int(getattr(cfg, "iter_max", 30)), int(nv > 1), _dp(T), _dp(UC), _dp(Sigma),
                                           _dp(kappa), 1))
                self.iter_stats = self.ctx.iter_stats()  # patches, flops and bytes of the CG steps, largest union (bench.py's roofline)
            else:
                # (no quality control, lakernel.py:774-777: T alone, the maps stay zero, A and -B/2 are never read)
                check(lib.imcom_solve_empir(self.ctx.handle, self.batch, _hp(self.n), self.ldn, self.m, None if nqc else _dp(self.A), None if nqc else _dp(mB),
                                            _hp(Cs), float(self.kappaC[0]), _dp(yx), _dp(self.y), _dp(self.x), float(cfg.rho),
                                            1 if nqc else 0, _dp(T), _dp(UC), _dp(Sigma), _dp(kappa), 1))
            Tt.zero_()
            Tt[:, :, : self.m] = T.transpose(1, 2)
        elif cfg.kernel != "Cholesky":
            raise NotImplementedError(f"resident path: no {cfg.kernel} kernel")
        elif cfg.fade == 0 and os.environ.get("IMCOM_EPILOGUE_FUSED") == "1":
            check(lib.imcom_solve_chol_resident_coadd(self.ctx.handle, self.batch, _hp(self.n), self.ldn, self.m, self.ldm,
                                                      _dp(self.A), _dp(Bt), _hp(Cs), _hp(self.kappaC), len(self.kappaC),
                                                      float(cfg.uctarget), float(cfg.sigmamax), _dp(Tt), _dp(UC), _dp(Sigma), _dp(kappa), _hp(info),
                                                      self.n2f, 0, cfg.n2, _dp(self.indata), cfg.n_inframe, _dp(self.expo), self.n_expo,
                                                      _dp(self.outimage_o[o]), _dp(self.Tsum_stamp_o[o]), _dp(self.Tsum_inpix_o[o]), _dp(self.Neff_o[o])))
            self._coadded.add(o)
        else:
            check(lib.imcom_solve_chol_resident(self.ctx.handle, self.batch, _hp(self.n), self.ldn, self.m, self.ldm,
                                            _dp(self.A), _dp(Bt), _hp(Cs), _hp(self.kappaC), len(self.kappaC),
                                            float(cfg.uctarget), float(cfg.sigmamax), _dp(Tt), _dp(UC),
                                            _dp(Sigma), _dp(kappa), _hp(info)))
        if cfg.kernel == "Iterative":  # coadd.py:1104-1107: "these could be negative as the iterative kernel is not exact"
            for t in (UC, Sigma):
                check(lib.imcom_clamp_min_f32(self.ctx.handle, _dp(t), t.numel(), 1e-32))
        if cfg.fade > 0 and not self._no_qlt_ctrl():  # (without quality control _build_system_matrices returns before the map tapers, coadd.py:1020-1025)
            for t in (kappa, Sigma, UC):
                check(lib.imcom_trapezoid_f32(self.ctx.handle, _dp(t), self.batch, self.n2f, cfg.fade))

    def coadd(self, only=None):
        """OutStamp._perform_coaddition (coadd.py:1294-1363).  ``only``: boolean mask [batch] -- the stamps to coadd (the call tapers T in
        place when fade > 0, so a stamp must be coadded once per solve: ``solve_end()`` says which stamps it solved again)."""
        self._stream()
        cfg = self.cfg
        done, self._coadded = getattr(self, "_coadded", set()), set()
        spans = [(0, self.batch)] if only is None else _runs(only)
        for o in range(self.n_out):
            if o in done:  # coadded inside solve()
                continue
            for s0, s1 in spans:
                check(lib.imcom_coadd_epilogue(self.ctx.handle, s1 - s0, _hp(self.n[s0:s1]), self.ldn, self.m, self.ldm, self.n2f,
                                               cfg.fade, cfg.n2, _dp(self.Tt_o[o][s0:]), _dp(self.indata[s0:]), cfg.n_inframe, _dp(self.expo[s0:]),
                                               self.n_expo, _dp(self.outimage_o[o][s0:]), _dp(self.Tsum_stamp_o[o][s0:]), _dp(self.Tsum_inpix_o[o][s0:]),
                                               _dp(self.Neff_o[o][s0:])))

    repair_share = 0.0   # share of the batch's stamps that took the Cholesky repair in its last solve (solve_begin / solve_end)
    EXPECT_REPAIR = 0.75  # a driver skips the doomed first factorisation of a pass when the pass before it was above this

    def run(self):
        self.build()
        self.solve_begin(expect_repair=self.repair_share >= self.EXPECT_REPAIR, repair_hint=getattr(self, "repair_absmax", None))
        if getattr(self, "_deferred", False):
            # the coaddition queued behind the solve's launches before the host waits for them (no gap between the two on the device);
            # a batch whose factorisation failed has been solved again by solve_end(): coadd its repaired T
            self.coadd()
            again = self.solve_end()
            if again is not False:
                self.coadd(only=again)
        else:
            self.solve_end()  # (kernels without the two halves: the synchronous solve)
            self.coadd()
        return self.result()

    def result(self, o=0):
        """Outputs for target PSF o (the reference's leading n_out axis, one StampBatchResult per target)."""
        s2 = (self.batch, self.n2f, self.n2f)
        return StampBatchResult(self.Tt_o[o], self.UC_o[o].view(s2), self.Sigma_o[o].view(s2), self.kappa_o[o].view(s2),
                                self.outimage_o[o].view(self.batch, self.cfg.n_inframe, self.n2f, self.n2f), self.Tsum_stamp_o[o],
                                self.Tsum_inpix_o[o].view(s2), self.Neff_o[o].view(s2), self.info_o[o].copy(), self.n.copy(), getattr(self, "perm", None))

    def results(self):
        return [self.result(o) for o in range(self.n_out)]
