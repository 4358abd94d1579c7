"""Input-pixel selection on the device (reference ``OutStamp._process_input_stamps`` coadd.py:886-977 with
``InStamp.make_selection`` 716-749): the block's InStamps are uploaded once as a pool, every output stamp then
gathers its nine neighbours' pixels inside the acceptance region straight into the StampBatch layouts."""

import ctypes as C

import numpy as np

from ._lib import MEM_DEVICE, check, default_context, lib


class InStampPool:
    """The InStamps of a block back to back on the GPU.

    ``instamps``: sequence of (x_val f64 [k], y_val f64 [k], data f32 [n_inframe, k], pix_cumsum int [n_expo+1]) --
    the attributes of the reference's ``InStamp`` (coadd.py:682-714)."""

    def __init__(self, instamps, n_inframe, device="cuda:0"):
        import torch

        self.n_inst = len(instamps)
        self.n_inframe = int(n_inframe)
        counts = [len(t[0]) for t in instamps]
        self.inst_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        npool = int(self.inst_off[-1])
        x, y = np.zeros(npool), np.zeros(npool)
        data = np.zeros((self.n_inframe, npool), np.float32)
        expo = np.zeros(npool, np.int32)
        for i, (xv, yv, dv, cum) in enumerate(instamps):
            o0, o1 = self.inst_off[i], self.inst_off[i + 1]
            x[o0:o1], y[o0:o1], data[:, o0:o1] = xv, yv, dv
            expo[o0:o1] = np.repeat(np.arange(len(cum) - 1), np.diff(cum))
        dev = torch.device(device)
        self.device = dev
        self.npool = npool
        # (through page-locked staging, queued without a host wait: five pageable copies of a 48 x 48-stamp block's pool -- 15 MB each for
        # x and y -- took 30 ms apiece with the device idle)
        from .stamps import h2d

        self.x, self.y = h2d(x, dev), h2d(y, dev)
        self.data, self.expo = h2d(data, dev), h2d(expo, dev)
        self.inst_off_dev = h2d(self.inst_off, dev)


def select_pixels(pool, inst_id, pivot_x, pivot_y, radius, ldn, ctx=None):
    """Selections of a batch of output stamps.  ``inst_id`` int [B,9] (-1 = absent), ``pivot_x/pivot_y`` float [B,9]
    (NaN = None), ``radius`` = rpix_search.  Returns device tensors x, y [B,ldn], indata [B,n_inframe,ldn],
    expo [B,ldn] and the host array cumsum [B,10] (inpix_cumsum of each stamp)."""
    import torch

    ctx = ctx if ctx is not None else default_context(pool.device.index or 0)
    inst_id = np.ascontiguousarray(inst_id, dtype=np.int32)
    B = inst_id.shape[0]
    assert inst_id.shape == (B, 9)
    dev = pool.device
    from .stamps import h2d  # (page-locked staging, no host wait: a pageable copy took 8 ms each beside a busy device)

    iid, pvx, pvy = h2d(inst_id, dev), h2d(pivot_x, dev, np.float64), h2d(pivot_y, dev, np.float64)
    x = torch.empty((B, ldn), dtype=torch.float64, device=dev)
    y = torch.empty_like(x)
    indata = torch.empty((B, pool.n_inframe, ldn), dtype=torch.float32, device=dev)
    expo = torch.empty((B, ldn), dtype=torch.int32, device=dev)
    cumsum = torch.empty((B, 10), dtype=torch.int32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    check(lib.imcom_select_pixels(ctx.handle, B, p(pool.x), p(pool.y), p(pool.data), pool.npool, pool.n_inframe, p(pool.expo),
                                  p(pool.inst_off_dev), pool.n_inst, p(iid), p(pvx), p(pvy), float(radius), int(ldn), p(x), p(y),
                                  p(indata), p(expo), p(cumsum), MEM_DEVICE))
    return x, y, indata, expo, cumsum.cpu().numpy()


def visiting_order(relevant_matrix, sp_arr):
    """Input-image indices (y, x) of the pixels InImage.partition_pixels visits, in its order (coadd.py:329-336):
    relevant sparse-grid cells row-major, pixels row-major inside a cell."""
    ys, xs = [], []
    sp_res = relevant_matrix.shape[0]
    for j_sp in range(sp_res):
        for i_sp in range(sp_res):
            if not relevant_matrix[j_sp, i_sp]:
                continue
            yy, xx = np.meshgrid(np.arange(sp_arr[j_sp], sp_arr[j_sp + 1]), np.arange(sp_arr[i_sp], sp_arr[i_sp + 1]), indexing="ij")
            ys.append(yy.ravel())
            xs.append(xx.ravel())
    cat = lambda p: np.concatenate(p).astype(np.uint16) if p else np.zeros(0, np.uint16)  # noqa: E731
    return cat(ys), cat(xs)


def partition_pixels(out_x, out_y, in_x, in_y, mask, use_instamps, n2, n1P, npixmax, device="cuda:0", ctx=None):
    """The binning of ``InImage.partition_pixels`` (coadd.py:329-358) on the device.  Inputs in visiting order (numpy or
    CUDA tensors); returns y_idx, x_idx (uint16), y_val, x_val (float64) [nst, nst, npixmax] and pix_count (uint32)
    [nst, nst] as CUDA tensors, nst = n1P + 2."""
    import torch

    dev = torch.device(device)
    ctx = ctx if ctx is not None else default_context(dev.index or 0)
    t = lambda a, dt: (a if torch.is_tensor(a) else torch.as_tensor(np.ascontiguousarray(a), device=dev)).to(dt).contiguous()  # noqa: E731
    ox, oy = t(out_x, torch.float64), t(out_y, torch.float64)
    ix, iy = t(in_x, torch.uint16), t(in_y, torch.uint16)
    mk = None if mask is None else t(mask, torch.uint8)
    nst = n1P + 2
    use = t(np.asarray(use_instamps).astype(np.uint8) if not torch.is_tensor(use_instamps) else use_instamps, torch.uint8)
    assert tuple(use.shape) == (nst, nst)
    npix = ox.numel()
    y_idx = torch.zeros((nst, nst, npixmax), dtype=torch.uint16, device=dev)
    x_idx = torch.zeros_like(y_idx)
    y_val = torch.zeros((nst, nst, npixmax), dtype=torch.float64, device=dev)
    x_val = torch.zeros_like(y_val)
    count = torch.zeros((nst, nst), dtype=torch.uint32, device=dev)
    pix_lower, pix_upper = -n2 - 0.5, n1P * n2 + n2 - 0.5  # coadd.py:208-209 (NsideP = n1P * n2)
    p = lambda a: None if a is None else C.c_void_p(a.data_ptr())  # noqa: E731
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    check(lib.imcom_partition_pixels(ctx.handle, npix, p(ox), p(oy), p(ix), p(iy), p(mk), p(use), nst, int(n2), pix_lower, pix_upper,
                                     int(npixmax), p(y_idx), p(x_idx), p(y_val), p(x_val), p(count)))
    return y_idx, x_idx, y_val, x_val, count
