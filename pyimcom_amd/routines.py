"""Native-routine seam: the reference's ``pyimcom_croutines`` function names on the HIP library.

Same names, argument order, in-place output semantics and ``None`` return as
``furry_parakeet.pyimcom_croutines`` / ``pyimcom.routine`` (reference src/pyimcom/routine.py:29-588,
imported at lakernel.py:41-47 and psfutil.py:37-49).  Arguments are C-contiguous float64 NumPy arrays;
the work runs on the GPU (host buffers are staged by the library).  No CPU fallback.
"""

import ctypes as C

import numpy as np

from . import _lib
from ._lib import MEM_HOST, check, default_context, lib


def _f64(a, name, writable=False):
    if not isinstance(a, np.ndarray) or a.dtype != np.float64 or not a.flags.c_contiguous:
        raise TypeError(f"{name} must be a C-contiguous float64 numpy array")
    if writable and not a.flags.writeable:
        raise ValueError(f"{name} must be writable")
    return a.ctypes.data_as(C.c_void_p)


def iD5512C_getw(w, fh):
    """routine.py:29-122 -- writes the 10 interpolation weights for offset ``fh`` into ``w``."""
    fh_a = np.array([fh], dtype=np.float64)
    out = np.zeros(10)
    check(lib.imcom_d5512_getw(default_context().handle, _f64(fh_a, "fh"), 1, _f64(out, "w"), MEM_HOST))
    w[:] = out


def iD5512C(infunc, xpos, ypos, fhatout):
    """routine.py:125-181 -- scattered 10x10 interpolation; off-grid outputs are left untouched."""
    nlayer, ngy, ngx = infunc.shape
    nout = xpos.size
    if ypos.size != nout or fhatout.shape != (nlayer, nout):
        raise ValueError("iD5512C: inconsistent shapes")
    check(lib.imcom_interp_d5512(default_context().handle, _f64(infunc, "infunc"), nlayer, ngy, ngx,
                                 _f64(xpos, "xpos"), _f64(ypos, "ypos"), nout, _f64(fhatout, "fhatout", True), 0,
                                 MEM_HOST))


def iD5512C_sym(infunc, xpos, ypos, fhatout):
    """routine.py:184-253 -- as iD5512C for a symmetric sqrt(nout) x sqrt(nout) output."""
    nlayer, ngy, ngx = infunc.shape
    nout = xpos.size
    if ypos.size != nout or fhatout.shape != (nlayer, nout):
        raise ValueError("iD5512C_sym: inconsistent shapes")
    check(lib.imcom_interp_d5512(default_context().handle, _f64(infunc, "infunc"), nlayer, ngy, ngx,
                                 _f64(xpos, "xpos"), _f64(ypos, "ypos"), nout, _f64(fhatout, "fhatout", True), 1,
                                 MEM_HOST))


def gridD5512C(infunc, xpos, ypos, fhatout):
    """routine.py:256-338 -- separable-grid interpolation, fhatout[npi, nyo*nxo]."""
    ngy, ngx = infunc.shape
    npi, nxo = xpos.shape[:2]
    nyo = ypos.shape[1]
    if ypos.shape[0] != npi or fhatout.shape != (npi, nyo * nxo):
        raise ValueError("gridD5512C: inconsistent shapes")
    check(lib.imcom_grid_d5512(default_context().handle, _f64(infunc, "infunc"), ngy, ngx, _f64(xpos, "xpos"),
                               _f64(ypos, "ypos"), npi, nxo, nyo, _f64(fhatout, "fhatout", True), MEM_HOST))


def lakernel1(lam, Q, mPhalf, C_, targetleak, kCmin, kCmax, nbis, kappa, Sigma, UC, T, smax):
    """routine.py:341-430 -- per-pixel kappa bisection (Q is unused, as in the reference)."""
    m, n = mPhalf.shape
    if lam.shape != (n,) or kappa.shape != (m,) or Sigma.shape != (m,) or UC.shape != (m,) or T.shape != (m, n):
        raise ValueError("lakernel1: inconsistent shapes")
    # float32 views are what lakernel.py:216-218 passes for kappa/Sigma/UC: stage through float64
    outs = []
    for a, name in ((kappa, "kappa"), (Sigma, "Sigma"), (UC, "UC")):
        outs.append(a if a.dtype == np.float64 and a.flags.c_contiguous else np.zeros(m))
    check(lib.imcom_lakernel1(default_context().handle, _f64(lam, "lam"), _f64(mPhalf, "mPhalf"), m, n, float(C_),
                              float(targetleak), float(kCmin), float(kCmax), int(nbis), _f64(outs[0], "kappa", True),
                              _f64(outs[1], "Sigma", True), _f64(outs[2], "UC", True), _f64(T, "T", True),
                              float(smax), MEM_HOST))
    for a, o in zip((kappa, Sigma, UC), outs):
        if a is not o:
            a[:] = o


def build_reduced_T_wrap(Nflat, Dflat, Eflat, kappa, ucmin, smax, out_kappa, out_Sigma, out_UC, out_w):
    """routine.py:487-588 -- reduced-space kappa search for the multi-kappa Cholesky path."""
    kappa = np.ascontiguousarray(kappa, dtype=np.float64)
    nv = kappa.size
    m = out_kappa.size
    if Nflat.size != m * nv * nv or Eflat.size != m * nv * nv or Dflat.size != m * nv or out_w.size != m * nv:
        raise ValueError("build_reduced_T_wrap: inconsistent shapes")
    Nflat, Dflat, Eflat = (np.ascontiguousarray(a, dtype=np.float64) for a in (Nflat, Dflat, Eflat))
    check(lib.imcom_build_reduced_T(default_context().handle, _f64(Nflat, "Nflat"), _f64(Dflat, "Dflat"),
                                    _f64(Eflat, "Eflat"), _f64(kappa, "kappa"), nv, m, float(ucmin), float(smax),
                                    _f64(out_kappa, "out_kappa", True), _f64(out_Sigma, "out_Sigma", True),
                                    _f64(out_UC, "out_UC", True), _f64(out_w, "out_w", True), MEM_HOST))


__all__ = ["iD5512C_getw", "iD5512C", "iD5512C_sym", "gridD5512C", "lakernel1", "build_reduced_T_wrap"]
