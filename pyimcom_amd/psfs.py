"""PSF sampling and target PSFs on the device: the host mirror of the reference's ``OutPSF`` static methods and of
``PSFGrp._get_outpsf / _sample_psf`` (src/pyimcom/psfutil.py:117-223, 709-795, 854-929).

Arrays may be numpy (host in, host out) or torch CUDA tensors (device in, device out)."""

import ctypes as C

import numpy as np

from ._lib import MEM_DEVICE, MEM_HOST, check, default_context, lib

QFILTER_NATIVE = [1.155, 1.456, 1.250, 1.021, 0.834, 0.689, 0.491, 1.009, 0.000, 1.159, 1.685]  # config.py:85-98
OBSC = 0.31


def _is_torch(a):
    return a is not None and type(a).__module__.startswith("torch")


def _out(shape, like, device):
    if like == "torch":
        import torch

        return torch.empty(shape, dtype=torch.float64, device=device)
    return np.empty(shape, dtype=np.float64)


def _p(a):
    if a is None:
        return None
    return C.c_void_p(a.data_ptr()) if _is_torch(a) else a.ctypes.data_as(C.c_void_p)


def _bind(ctx, device):
    """Device-mode calls enqueue on the context's stream and return without a sync: that stream must be torch's current
    one, or torch could consume the output tensor before the kernel has written it."""
    if device is not None:
        import torch

        ctx.set_stream(torch.cuda.current_stream(torch.device(device)).cuda_stream)


def psf_gaussian(n, sigmax, sigmay, device=None, ctx=None):
    """``OutPSF.psf_gaussian`` (psfutil.py:117-146); ``device`` given -> torch tensor on that device."""
    ctx = ctx or default_context()
    out = _out((n, n), "torch" if device else "numpy", device)
    _bind(ctx, device)
    check(lib.imcom_psf_gaussian(ctx.handle, int(n), float(sigmax), float(sigmay), _p(out), MEM_DEVICE if device else MEM_HOST))
    return out


def psf_simple_airy(n, ldp, obsc=0.0, tophat_conv=0.0, sigma=0.0, device=None, ctx=None):
    """``OutPSF.psf_simple_airy`` (psfutil.py:148-223)."""
    ctx = ctx or default_context()
    out = _out((n, n), "torch" if device else "numpy", device)
    _bind(ctx, device)
    check(lib.imcom_psf_simple_airy(ctx.handle, int(n), float(ldp), float(obsc), float(tophat_conv), float(sigma), _p(out),
                                    MEM_DEVICE if device else MEM_HOST))
    return out


def smooth_and_pad(inArray, tophatwidth=0.0, gaussiansigma=0.0, ctx=None):
    """``InImage.smooth_and_pad`` (coadd.py:433-474): smear a PSF image [ny, nx] (or a stack [n, ny, nx]) with a top-hat
    and a Gaussian, padded by ``npad`` on every side.  numpy in -> numpy out; torch (device) in -> torch out."""
    ctx = ctx or default_context()
    single = inArray.ndim == 2
    a = inArray[None] if single else inArray
    n, ny, nx = a.shape
    npad = int(lib.imcom_smooth_pad_width(float(tophatwidth), float(gaussiansigma)))
    if _is_torch(a):
        import torch

        a = a.to(torch.float64).contiguous()
        out = torch.empty((n, ny + 2 * npad, nx + 2 * npad), dtype=torch.float64, device=a.device)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        mem = MEM_DEVICE
    else:
        a = np.ascontiguousarray(a, dtype=np.float64)
        out = np.empty((n, ny + 2 * npad, nx + 2 * npad))
        mem = MEM_HOST
    check(lib.imcom_smooth_and_pad(ctx.handle, n, _p(a), ny, nx, float(tophatwidth), float(gaussiansigma), _p(out), mem))
    return out[0] if single else out


def get_outpsf(outpsf, extrasmooth, use_filter, nsamp, oversamp, device=None, ctx=None):
    """``PSFGrp._get_outpsf`` (psfutil.py:854-896): the (nsamp+1)^2 target PSF image."""
    if outpsf == "GAUSSIAN":
        return psf_gaussian(nsamp + 1, extrasmooth * oversamp, extrasmooth * oversamp, device, ctx)
    if outpsf in ("AIRYOBSC", "AIRYUNOBSC"):
        return psf_simple_airy(nsamp + 1, QFILTER_NATIVE[use_filter] * oversamp, OBSC if outpsf == "AIRYOBSC" else 0.0, 0.0,
                               extrasmooth * oversamp, device, ctx)
    raise RuntimeError("Error: unsupported target output PSF type")


def sample_psf(psf, nsamp, yxco=None, psf_circ=False, psf_norm=False, ctx=None):
    """``PSFGrp._sample_psf`` + the cut-out / normalisation of ``PSFGrp.__init__`` (psfutil.py:709-795, 650-656).

    psf [n_psf, ny, nx]; yxco [n_psf, 2, nsamp, nsamp] (y, x offsets from the image centre) or None for the
    unrotated grid.  Returns psf_arr [n_psf, nsamp, nsamp]."""
    ctx = ctx or default_context()
    tor = _is_torch(psf)
    if tor:
        import torch

        psf = psf.contiguous()
        yxco = None if yxco is None else yxco.contiguous()
        ctx.set_stream(torch.cuda.current_stream(psf.device).cuda_stream)
    else:
        psf = np.ascontiguousarray(psf, dtype=np.float64)
        yxco = None if yxco is None else np.ascontiguousarray(yxco, dtype=np.float64)
    n_psf, ny, nx = psf.shape
    if yxco is not None:
        assert tuple(yxco.shape) == (n_psf, 2, nsamp, nsamp)
    out = _out((n_psf, nsamp, nsamp), "torch" if tor else "numpy", psf.device if tor else None)
    check(lib.imcom_sample_psf(ctx.handle, n_psf, _p(psf), ny, nx, _p(yxco), int(nsamp), int(bool(psf_circ)), int(bool(psf_norm)),
                               _p(out), MEM_DEVICE if tor else MEM_HOST))
    return out


def lattice_nodes_and_weights(grid, L=17):
    """Chebyshev-Lobatto nodes u_a over [grid[0], grid[-1]] and W [len(grid), L], W[i][a] = the a-th Lagrange basis polynomial of the
    nodes at grid[i] (barycentric form, weights (-1)^a, halved at the ends) -- the host inputs of ``lattice_positions``."""
    grid = np.asarray(grid, dtype=np.float64)
    mid, half = 0.5 * (grid[0] + grid[-1]), 0.5 * (grid[-1] - grid[0])
    a = np.arange(L)
    nodes = mid + half * np.cos(np.pi * a / (L - 1))
    w = (-1.0) ** a
    w[0] *= 0.5
    w[-1] *= 0.5
    d = grid[:, None] - nodes[None, :]
    hit = d == 0.0
    d[hit] = 1.0
    W = w[None, :] / d
    W /= W.sum(axis=1, keepdims=True)
    rows = hit.any(axis=1)
    W[rows] = hit[rows].astype(np.float64)
    return nodes, np.ascontiguousarray(W)


def lattice_positions(lattice, W, nsamp, ctx=None):
    """The sampling positions yxco [count, 2, nsamp, nsamp] of ``sample_psf`` from their values on an L x L lattice of Chebyshev-Lobatto
    nodes (imcom_lattice_positions): ``lattice`` [count, 2, L, L] (numpy -> numpy out, torch device tensor -> device out), ``W`` from
    ``lattice_nodes_and_weights``.  Replaces the nsamp^2 WCS evaluations per PSF group and exposure of psfutil.py:751-771."""
    ctx = ctx or default_context()
    tor = _is_torch(lattice)
    count, two, L, L2 = lattice.shape
    assert two == 2 and L == L2 and W.shape == (nsamp, L)
    W = np.ascontiguousarray(W, dtype=np.float64)
    if tor:
        import torch

        lattice = lattice.contiguous()
        ctx.set_stream(torch.cuda.current_stream(lattice.device).cuda_stream)
    else:
        lattice = np.ascontiguousarray(lattice, dtype=np.float64)
    out = _out((count, 2, nsamp, nsamp), "torch" if tor else "numpy", lattice.device if tor else None)
    check(lib.imcom_lattice_positions(ctx.handle, int(count), int(L), _p(W), _p(lattice), int(nsamp), _p(out), MEM_DEVICE if tor else MEM_HOST))
    return out
