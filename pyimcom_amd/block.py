"""Device-side block maps: the accumulators of Block.coadd_output_stamps /
_output_stamp_wrapper (reference src/pyimcom/coadd.py:2031-2047, 1975-1993) kept on the GPU, and the
boundary recovery of Block.build_output_file (coadd.py:2163-2181)."""

import ctypes as C

import numpy as np
import torch

from ._lib import check, default_context, lib


def _dp(t):
    return C.c_void_p(t.data_ptr())


def _hp(a):
    return a.ctypes.data_as(C.c_void_p)


class BlockMaps:
    """The reference's block arrays (coadd.py:2031-2047), float32: out_map [n_out, n_inframe, NsidePf, NsidePf], the
    UC / Sigma / kappa / Tsum / Neff maps [n_out, NsidePf, NsidePf], T_weightmap [n_out, n_expo, n1P, n1P].

    With fade > 0 neighbouring stamps overlap and a float32 sum depends on its order.  The stamps' tiles are therefore kept
    apart in four index-parity LAYERS (stamps of equal parity never overlap; ``layers[name]``: [n_out, 4, ...] in the dtype
    the tiles arrive in) and added per pixel in the order of the reference's loop when the maps are read -- ``order="cells"``:
    coadd.py:2056-2059, cells of 2 x 2 stamps from ``origin`` = (j_st_min, i_st_min) on, row by row of cells, inside a cell
    dj outer, di inner; ``order="rows"``: j_st outer, i_st inner -- so ``out_map`` / ``maps`` are the same bits whatever batches,
    passes or processes the stamps were dealt to, and the reference's own rounding.  With fade == 0 every pixel has one
    contribution and the tiles go straight to the maps."""

    NAMES = ("UC", "Sigma", "kappa", "Tsum", "Neff")

    def __init__(self, n1P, n2, fade, n_inframe, n_expo, ctx=None, device="cuda:0", n_out=1, order="cells", origin=(1, 1)):
        self.n1P, self.n2, self.fade, self.n_inframe, self.n_expo, self.n_out = n1P, n2, fade, n_inframe, n_expo, n_out
        assert order in ("cells", "rows")
        self.order, self.origin = order, (int(origin[0]), int(origin[1]))
        self.nside = n1P * n2 + 2 * fade  # NsidePf (coadd.py:2029)
        self.ctx = ctx or default_context()
        self.device = dev = torch.device(device)
        f32 = torch.float32
        self._out_map = torch.zeros((n_out, n_inframe, self.nside, self.nside), dtype=f32, device=dev)
        self._maps = {k: torch.zeros((n_out, self.nside, self.nside), dtype=f32, device=dev) for k in self.NAMES}
        self.T_weightmap = torch.zeros((n_out, n_expo, n1P, n1P), dtype=f32, device=dev)
        self.layers = {}  # name -> [n_out, 4, nlayer, nside, nside], allocated when the first tile of that dtype arrives (fade > 0)
        self._dirty = self._recovered = False

    # the maps as the reference's loop leaves them (read access combines the layers first)
    @property
    def out_map(self):
        self.combine()
        return self._out_map

    @property
    def maps(self):
        self.combine()
        return self._maps

    def _layer(self, name, nlayer, dtype):
        t = self.layers.get(name)
        if t is None:
            t = self.layers[name] = torch.zeros((self.n_out, 4, nlayer, self.nside, self.nside), dtype=dtype, device=self.device)
        assert t.dtype == dtype, f"tiles of {name} changed their dtype"
        return t

    def add(self, res, jst, ist):
        """Add a finished batch: a StampBatchResult, or the list of n_out of them (StampBatch.results());
        jst/ist = 1-based OutStamp indices of its stamps (coadd.py:1963)."""
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        jst = np.ascontiguousarray(jst, dtype=np.int32)
        ist = np.ascontiguousarray(ist, dtype=np.int32)
        b = len(jst)
        h = self.ctx.handle

        def acc(src, nlayer, name, o):
            src = src.contiguous()
            f64 = 1 if src.dtype == torch.float64 else 0
            if self.fade == 0:  # no overlap: one contribution per pixel
                dst = self._out_map[o] if name == "out_map" else self._maps[name][o]
                check(lib.imcom_block_accumulate(h, b, _hp(jst), _hp(ist), self.n2, self.fade, nlayer, _dp(src), f64, _dp(dst), self.nside))
            else:
                check(lib.imcom_block_place(h, b, _hp(jst), _hp(ist), self.n2, self.fade, nlayer, _dp(src), f64,
                                            _dp(self._layer(name, nlayer, src.dtype)[o]), self.nside))

        results = list(res) if isinstance(res, (list, tuple)) else [res]
        assert len(results) == self.n_out
        up = lambda a: torch.from_numpy(a).pin_memory().to(self.T_weightmap.device, non_blocking=True)  # noqa: E731  (no stream drain)
        jj, ii = up(jst.astype(np.int64) - 1), up(ist.astype(np.int64) - 1)
        for o, r in enumerate(results):
            acc(r.outimage, self.n_inframe, "out_map", o)
            acc(r.UC, 1, "UC", o)
            acc(r.Sigma, 1, "Sigma", o)
            acc(r.kappa, 1, "kappa", o)
            acc(r.Tsum_inpix, 1, "Tsum", o)
            acc(r.Neff, 1, "Neff", o)
            # T_weightmap[:, j_st-1, i_st-1] = Tsum_stamp (coadd.py:1981): plain indexed copy
            self.T_weightmap[o][:, jj, ii] = r.Tsum_stamp[:, : self.n_expo].T.to(torch.float32)
        self._dirty = self.fade > 0

    def combine(self):
        """Maps <- the layers' sums in the reference's stamp order (a no-op with fade == 0 or when nothing was added since)."""
        if not self._dirty:
            return
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        for name, lay in self.layers.items():
            for o in range(self.n_out):
                dst = self._out_map[o] if name == "out_map" else self._maps[name][o]
                check(lib.imcom_block_combine(self.ctx.handle, self.n1P, self.n2, self.fade, lay.shape[2], _dp(lay[o]),
                                              1 if lay.dtype == torch.float64 else 0, _dp(dst), self.nside, 1 if self.order == "cells" else 0,
                                              self.origin[0], self.origin[1]))
        self._dirty, self._recovered = False, False

    def state(self):
        """What a process has coadded of the block, as host arrays that ADD exactly between processes (every stamp of a block
        is coadded by one of them): the parity layers (fade > 0; keys ``L_<name>``) or the maps themselves, and T_weightmap."""
        src = {f"L_{k}": v for k, v in self.layers.items()} if self.fade > 0 else dict(self._maps, out_map=self._out_map)
        out = {k: v.cpu().numpy() for k, v in src.items()}
        out["T_weightmap"] = self.T_weightmap.cpu().numpy()
        return out

    def load_state(self, arrays):
        """Inverse of state() (arrays: the sum of the processes' states)."""
        for k, v in arrays.items():
            t = torch.as_tensor(np.asarray(v))
            if k.startswith("L_"):
                self._layer(k[2:], t.shape[2], t.dtype).copy_(t)
            elif k == "T_weightmap":
                self.T_weightmap.copy_(t)
            elif k == "out_map":
                self._out_map.copy_(t)
            elif k in self._maps:
                self._maps[k].copy_(t)
        self._dirty = self.fade > 0

    def arrays(self):
        """The block's maps as host arrays (names of farm.write_block)."""
        out = {"out_map": self.out_map.cpu().numpy(), "T_weightmap": self.T_weightmap.cpu().numpy()}
        out.update({k: v.cpu().numpy() for k, v in self.maps.items()})
        return out

    COMPRESS = {"UC": (-5000, True), "Sigma": (-10000, False), "kappa": (-5000, True), "Tsum": (200000, False),
                "Neff": (50000, True)}  # coefficient, unsigned (coadd.py:2249-2303)

    def compress(self, name, fk=0):
        """Block.compress_map (coadd.py:2087-2138) of one quality map, cropped by fk on every side as
        build_output_file does: (u)int16 tensor of coef * log10(map)."""
        coef, uns = self.COMPRESS[name]
        m = self.maps[name][:, fk : self.nside - fk, fk : self.nside - fk].contiguous()
        out = torch.empty(m.shape, dtype=torch.uint16 if uns else torch.int16, device=m.device)
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        check(lib.imcom_compress_map_f32(self.ctx.handle, _dp(m), m.numel(), coef, 1 if uns else 0, _dp(out)))
        return out

    def finalize(self, pad_sides="", postage_pad=0):
        """coadd.py:2163-2181: recover the faded block boundary (the padding sides listed in `pad_sides` are
        recovered at the array edge, the others `postage_pad` stamps further in)."""
        self.combine()
        assert not self._recovered, "the block boundary has been recovered already"
        self._recovered = True
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        h = self.ctx.handle
        check(lib.imcom_trapezoid_recover_f32(h, _dp(self.out_map), self.n_out * self.n_inframe, self.nside, self.nside, self.fade, 0, 0, 0, 0))
        w = postage_pad * self.n2
        pads = [w * (s not in pad_sides) for s in "BTLR"]
        for m in self.maps.values():
            check(lib.imcom_trapezoid_recover_f32(h, _dp(m), self.n_out, self.nside, self.nside, self.fade, *pads))
