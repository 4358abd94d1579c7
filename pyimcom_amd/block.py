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
    UC / Sigma / kappa / Tsum / Neff maps [n_out, NsidePf, NsidePf], T_weightmap [n_out, n_expo, n1P, n1P]."""

    def __init__(self, n1P, n2, fade, n_inframe, n_expo, ctx=None, device="cuda:0", n_out=1):
        self.n1P, self.n2, self.fade, self.n_inframe, self.n_expo, self.n_out = n1P, n2, fade, n_inframe, n_expo, n_out
        self.nside = n1P * n2 + 2 * fade  # NsidePf (coadd.py:2029)
        self.ctx = ctx or default_context()
        dev = torch.device(device)
        f32 = torch.float32
        self.out_map = torch.zeros((n_out, n_inframe, self.nside, self.nside), dtype=f32, device=dev)
        self.maps = {k: torch.zeros((n_out, self.nside, self.nside), dtype=f32, device=dev) for k in ("UC", "Sigma", "kappa", "Tsum", "Neff")}
        self.T_weightmap = torch.zeros((n_out, n_expo, n1P, n1P), dtype=f32, device=dev)

    def add(self, res, jst, ist):
        """Add a finished batch: a StampBatchResult, or the list of n_out of them (StampBatch.results());
        jst/ist = 1-based OutStamp indices of its stamps (coadd.py:1963)."""
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        jst = np.ascontiguousarray(jst, dtype=np.int32)
        ist = np.ascontiguousarray(ist, dtype=np.int32)
        b = len(jst)
        h = self.ctx.handle

        def acc(src, nlayer, dst):
            src = src.contiguous()
            check(lib.imcom_block_accumulate(h, b, _hp(jst), _hp(ist), self.n2, self.fade, nlayer, _dp(src),
                                             1 if src.dtype == torch.float64 else 0, _dp(dst), self.nside))

        results = list(res) if isinstance(res, (list, tuple)) else [res]
        assert len(results) == self.n_out
        up = lambda a: torch.from_numpy(a).pin_memory().to(self.T_weightmap.device, non_blocking=True)  # noqa: E731  (no stream drain)
        jj, ii = up(jst.astype(np.int64) - 1), up(ist.astype(np.int64) - 1)
        for o, r in enumerate(results):
            acc(r.outimage, self.n_inframe, self.out_map[o])
            acc(r.UC, 1, self.maps["UC"][o])
            acc(r.Sigma, 1, self.maps["Sigma"][o])
            acc(r.kappa, 1, self.maps["kappa"][o])
            acc(r.Tsum_inpix, 1, self.maps["Tsum"][o])
            acc(r.Neff, 1, self.maps["Neff"][o])
            # T_weightmap[:, j_st-1, i_st-1] = Tsum_stamp (coadd.py:1981): plain indexed copy
            self.T_weightmap[o][:, jj, ii] = r.Tsum_stamp[:, : self.n_expo].T.to(torch.float32)

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
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        h = self.ctx.handle
        check(lib.imcom_trapezoid_recover_f32(h, _dp(self.out_map), self.n_out * self.n_inframe, self.nside, self.nside, self.fade, 0, 0, 0, 0))
        w = postage_pad * self.n2
        pads = [w * (s not in pad_sides) for s in "BTLR"]
        for m in self.maps.values():
            check(lib.imcom_trapezoid_recover_f32(h, _dp(m), self.n_out, self.nside, self.nside, self.fade, *pads))
