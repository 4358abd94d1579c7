"""Kernel-class seam: drop-in linear-algebra kernels for ``OutStamp.LAKERNEL``.

Mirrors the contract of the reference's ``_LAKernel`` (src/pyimcom/lakernel.py:50-138, exercised by
tests/pyimcom/test_la.py:65-90): ``K(outst)`` then ``K()`` reads ``outst.sysmata`` (f64 [N,N], left
unmodified), ``outst.mhalfb`` (f64 [n_out,m,N]), ``outst.outovlc`` (f64 [n_out]),
``outst.inpix_cumsum[-1]`` and ``outst.blk.cfg.{n_out,n2f,kappaC_arr,uctarget,sigmamax}``; it writes
``outst.T`` (f32 [n_out,m,N]) and ``outst.UC/Sigma/kappa`` (f32 [n_out,n2f,n2f]).

Registering::

    from pyimcom.coadd import OutStamp
    from pyimcom_amd.lakernel import HipCholKernel, HipEigenKernel
    OutStamp.LAKERNEL["Cholesky"] = HipCholKernel     # or a new key selected by cfg.linear_algebra
    OutStamp.LAKERNEL["Eigen"] = HipEigenKernel

All arithmetic runs in libimcom_hip (fp64 MFMA); there is no CPU fallback.
"""

import ctypes as C

import numpy as np

from ._lib import MEM_HOST, check, default_context, lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _host_out(shape, dtype=np.float32, zero=False):
    """Output array for the device to write into: page-locked memory from torch's caching host allocator, seen as a numpy array
    (the block goes back to the cache when the array dies).  Into fresh pageable memory a 20 MB T arrives at ~16 GB/s (page
    faults under the driver's staging copy), and freeing such an array later costs the next call ~20 ms (the driver had
    registered its pages; unmapping them stops the process's GPU queues until they are revalidated)."""
    if int(np.prod(shape)) == 0:
        return np.zeros(shape, dtype=dtype)
    import torch

    make = torch.zeros if zero else torch.empty
    return make(tuple(int(v) for v in shape), dtype=getattr(torch, np.dtype(dtype).name), pin_memory=True).numpy()


class _repair_memory:
    """The reference's stamp loop calls the kernel class once per output stamp, neighbours one after the other (coadd.py:2056-2059).  Where
    _cholesky_wrapper's repair is the rule (lakernel.py:262-279; the reference's production configuration: DESIGN.md section 4) a stamp's
    smallest eigenvalue lies within a few per cent of its neighbours': the largest |w[0]| of the context's last sixteen repaired stamps is
    handed to the library as the next call's starting point (imcom_ctx_set_repair_hint: one factorisation inside the smallest-eigenvalue
    iteration instead of two; what the iteration converges to does not change) and the call's own repairs are remembered.  When the last
    two calls both ended in the repair, the next one is told to EXPECT it (imcom_ctx_set_repair_expect): it does not attempt the
    factorisation of A + kappa I that is known to fail (10 % of a production stamp's call); a stamp that needs no repair after all is
    recognised and solved plainly."""

    _recent = {}  # context handle -> deque of max |w[0]| per call
    _streak = {}  # context handle -> calls in a row that ended in the repair

    def __init__(self, ctx):
        self.ctx = ctx

    def __enter__(self):
        import collections

        self.q = self._recent.setdefault(id(self.ctx), collections.deque(maxlen=16))
        self.ctx.set_repair_hint(max(self.q) if self.q else 0.0)
        self.ctx.set_repair_expect(self._streak.get(id(self.ctx), 0) >= 2)
        return self

    def __exit__(self, exc_type, exc, tb):
        try:
            if exc_type is None:
                cnt, lo, hi = self.ctx.last_repair()
                if cnt:
                    self.q.append(max(abs(lo), abs(hi)))
                self._streak[id(self.ctx)] = self._streak.get(id(self.ctx), 0) + 1 if cnt else 0
        finally:
            if self.ctx._h:
                self.ctx.set_repair_hint(0.0)
                self.ctx.set_repair_expect(False)
        return False


class _HipLAKernel:
    """Shared constructor / output handling (lakernel.py:68-138)."""

    writes_all_of_T = False  # (the device call fills every entry of T: the output array need not be zeroed first)

    def __init__(self, outst, ctx=None):
        self.outst = outst
        cfg = outst.blk.cfg
        self.n_out = cfg.n_out
        self.m = cfg.n2f**2
        self.n = int(self.outst.inpix_cumsum[-1])
        self.n2f = cfg.n2f
        self.kappaC_arr = np.ascontiguousarray(np.atleast_1d(cfg.kappaC_arr), dtype=np.float64)
        self.nv = self.kappaC_arr.size
        self.ucmin = float(cfg.uctarget)
        self.smax = float(cfg.sigmamax)
        self.ctx = ctx if ctx is not None else default_context()
        self.info = None

    def _solve(self, A, B, C_, T, UC, Sigma, kappa, info):  # pragma: no cover - abstract
        raise NotImplementedError

    def __call__(self):
        shape = (self.n_out, self.n2f, self.n2f)
        n, m = self.n, self.m
        if n == 0:  # lakernel.py:110-119
            self.outst.T = np.zeros((self.n_out, m, 0), dtype=np.float32)
            self.outst.UC = np.ones(shape, dtype=np.float32)
            self.outst.Sigma = np.zeros(shape, dtype=np.float32)
            self.outst.kappa = np.ones(shape, dtype=np.float32)
            return
        A = np.ascontiguousarray(self.outst.sysmata, dtype=np.float64)
        mBhalf = np.ascontiguousarray(self.outst.mhalfb, dtype=np.float64)
        Cs = np.ascontiguousarray(np.atleast_1d(self.outst.outovlc), dtype=np.float64)
        if A.shape != (n, n) or mBhalf.shape != (self.n_out, m, n) or Cs.shape != (self.n_out,):
            raise ValueError("sysmata / mhalfb / outovlc shapes do not match the stamp")
        T = _host_out((self.n_out, m, n), zero=not self.writes_all_of_T)
        UC = np.zeros((self.n_out, m), dtype=np.float32)
        Sigma = np.zeros((self.n_out, m), dtype=np.float32)
        kappa = np.zeros((self.n_out, m), dtype=np.float32)
        self.info = np.zeros((self.n_out,), dtype=np.int32)
        # one factorisation per target PSF since kappa = kappaC * C[j_out] (lakernel.py:291-299, 349-353)
        for j in range(self.n_out):
            self._solve(A, mBhalf[j], Cs[j : j + 1], T[j], UC[j], Sigma[j], kappa[j], self.info[j : j + 1])
        self.outst.T = T
        self.outst.UC = UC.reshape(shape)
        self.outst.Sigma = Sigma.reshape(shape)
        self.outst.kappa = kappa.reshape(shape)


class HipCholKernel(_HipLAKernel):
    """Cholesky path: lakernel.CholKernel (lakernel.py:226-394), single- and multi-kappa."""

    writes_all_of_T = True

    def _solve(self, A, B, C_, T, UC, Sigma, kappa, info):
        n_arr = np.array([self.n], dtype=np.int32)
        with _repair_memory(self.ctx):
            check(lib.imcom_solve_chol(self.ctx.handle, 1, _ptr(n_arr), self.n, self.m, _ptr(A), _ptr(B), _ptr(C_),
                                       _ptr(self.kappaC_arr), self.nv, self.ucmin, self.smax, _ptr(T), _ptr(UC),
                                       _ptr(Sigma), _ptr(kappa), _ptr(info), MEM_HOST))


def solve_chol_stamps(outstamps, ctx=None):
    """``HipCholKernel(outst)()`` for several OutStamps in ONE call: the reference runs the kernel of one OutStamp at a time
    (coadd.py:1091-1093); a caller that holds the system matrices of several -- the four OutStamps of a 2 x 2 group share their
    PSF overlaps, coadd.py:839-844 -- hands them over together and gets one batched factorisation and solve, -B/2 crossing PCIe
    behind the factorisation.  Same outputs on every OutStamp as the single call (T, UC, Sigma, kappa; lakernel.py:122-138);
    returns the kernels (``info`` set)."""
    kernels = [HipCholKernel(o, ctx) for o in outstamps]
    if not kernels:
        return kernels
    k0 = kernels[0]
    for k in kernels:
        if k.m != k0.m or k.n_out != k0.n_out or not np.array_equal(k.kappaC_arr, k0.kappaC_arr) or k.ucmin != k0.ucmin or k.smax != k0.smax:
            raise ValueError("the OutStamps of one call must share the output grid, the kappa nodes and the targets")
    nst, m, n_out = len(kernels), k0.m, k0.n_out
    shape = (n_out, k0.n2f, k0.n2f)
    n_arr = np.array([k.n for k in kernels], dtype=np.int32)
    As, Bs, Cs = [], [], np.zeros((nst, n_out))
    for i, k in enumerate(kernels):
        A = np.ascontiguousarray(k.outst.sysmata, dtype=np.float64) if k.n else np.zeros((0, 0))
        B = np.ascontiguousarray(k.outst.mhalfb, dtype=np.float64) if k.n else np.zeros((n_out, m, 0))
        Cs[i] = np.atleast_1d(k.outst.outovlc)
        if A.shape != (k.n, k.n) or B.shape != (n_out, m, k.n):
            raise ValueError("sysmata / mhalfb shapes do not match the stamp")
        As.append(A)
        Bs.append(B)
    T = [_host_out((n_out, m, k.n)) for k in kernels]
    UC = np.zeros((nst, n_out, m), dtype=np.float32)
    Sigma = np.zeros((nst, n_out, m), dtype=np.float32)
    kappa = np.zeros((nst, n_out, m), dtype=np.float32)
    info = np.zeros((n_out, nst), dtype=np.int32)
    PP = C.c_void_p * nst

    def ptrs(arrs):
        return PP(*[a.ctypes.data for a in arrs])

    pA = ptrs(As)
    # one factorisation per target PSF since kappa = kappaC * C[j_out] (lakernel.py:291-299, 349-353)
    for j in range(n_out):
        Cj = np.ascontiguousarray(Cs[:, j])
        with _repair_memory(k0.ctx):
                check(lib.imcom_solve_chol_stamps(k0.ctx.handle, nst, _ptr(n_arr), m, pA, ptrs([b[j] for b in Bs]), _ptr(Cj), _ptr(k0.kappaC_arr), k0.nv,
                                              k0.ucmin, k0.smax, ptrs([t[j] for t in T]), ptrs([UC[i, j] for i in range(nst)]),
                                              ptrs([Sigma[i, j] for i in range(nst)]), ptrs([kappa[i, j] for i in range(nst)]), _ptr(info[j])))
    for i, k in enumerate(kernels):
        k.info = np.ascontiguousarray(info[:, i])
        k.outst.T = T[i]
        k.outst.UC = UC[i].reshape(shape)
        k.outst.Sigma = Sigma[i].reshape(shape)
        k.outst.kappa = kappa[i].reshape(shape)
    return kernels


class HipEigenKernel(_HipLAKernel):
    """Eigendecomposition path: lakernel.EigenKernel (lakernel.py:141-223); nbis as line 174."""

    nbis = 13
    writes_all_of_T = True

    def _solve(self, A, B, C_, T, UC, Sigma, kappa, info):
        n_arr = np.array([self.n], dtype=np.int32)
        check(lib.imcom_solve_eigen(self.ctx.handle, 1, _ptr(n_arr), self.n, self.m, _ptr(A), _ptr(B), _ptr(C_),
                                    _ptr(self.kappaC_arr), self.nv, self.ucmin, self.smax, int(self.nbis), _ptr(T),
                                    _ptr(UC), _ptr(Sigma), _ptr(kappa), _ptr(info), MEM_HOST))


class _HipGeomKernel(_HipLAKernel):
    """Kernels that use the pixel geometry (lakernel.py:617-622, 757-761): outst.yx_val, iny_val, inx_val and
    rho_acc = (cfg.instamp_pad / arcsec) / (cfg.dtheta * 3600)."""

    ARCSEC = np.pi / 180.0 / 3600.0  # pyimcom.config.Settings.arcsec (config.py:85-98)

    def _geometry(self):
        cfg = self.outst.blk.cfg
        yx = np.ascontiguousarray(np.asarray(self.outst.yx_val, dtype=np.float64).reshape(2, self.m))
        iny = np.ascontiguousarray(self.outst.iny_val, dtype=np.float64)
        inx = np.ascontiguousarray(self.outst.inx_val, dtype=np.float64)
        rho_acc = (cfg.instamp_pad / self.ARCSEC) / (cfg.dtheta * 3600.0)
        return yx, iny, inx, float(rho_acc)


class HipIterKernel(_HipGeomKernel):
    """Iterative path: lakernel.IterKernel (lakernel.py:533-744), restricted CG per output pixel.

    ``exact_UC`` None keeps the reference defaults (False for one kappa node, True for several)."""

    exact_UC = None

    def _solve(self, A, B, C_, T, UC, Sigma, kappa, info):
        cfg = self.outst.blk.cfg
        yx, iny, inx, rho = self._geometry()
        exact = (self.nv > 1) if self.exact_UC is None else bool(self.exact_UC)
        n_arr = np.array([self.n], dtype=np.int32)
        check(lib.imcom_solve_iter(self.ctx.handle, 1, _ptr(n_arr), self.n, self.m, _ptr(A), _ptr(B), _ptr(C_),
                                   _ptr(self.kappaC_arr), self.nv, self.ucmin, self.smax, _ptr(yx), _ptr(iny), _ptr(inx),
                                   rho, float(cfg.iter_rtol), int(cfg.iter_max), int(exact), _ptr(T), _ptr(UC),
                                   _ptr(Sigma), _ptr(kappa), MEM_HOST))


class HipEmpirKernel(_HipGeomKernel):
    """Empirical path: lakernel.EmpirKernel (lakernel.py:747-805); honours ``outst.no_qlt_ctrl``."""

    def __call__(self):
        if not getattr(self.outst, "no_qlt_ctrl", False) or self.n == 0:
            return super().__call__()
        # no quality control (coadd.py:1019-1024): the system matrices were never built; only T is produced
        shape = (self.n_out, self.n2f, self.n2f)
        yx, iny, inx, rho = self._geometry()
        T = np.zeros((self.n_out, self.m, self.n), dtype=np.float32)
        maps = np.zeros((3, self.m), dtype=np.float32)
        n_arr = np.array([self.n], dtype=np.int32)
        check(lib.imcom_solve_empir(self.ctx.handle, 1, _ptr(n_arr), self.n, self.m, None, None, None, 0.0, _ptr(yx),
                                    _ptr(iny), _ptr(inx), rho, 1, _ptr(T[0]), _ptr(maps[0]), _ptr(maps[1]), _ptr(maps[2]),
                                    MEM_HOST))
        T[1:] = T[0]  # the same weights for every target PSF (lakernel.py:775)
        self.outst.T = T
        self.outst.UC = np.zeros(shape, dtype=np.float32)
        self.outst.Sigma = np.zeros(shape, dtype=np.float32)
        self.outst.kappa = np.zeros(shape, dtype=np.float32)

    def _solve(self, A, B, C_, T, UC, Sigma, kappa, info):
        yx, iny, inx, rho = self._geometry()
        n_arr = np.array([self.n], dtype=np.int32)
        check(lib.imcom_solve_empir(self.ctx.handle, 1, _ptr(n_arr), self.n, self.m, _ptr(A), _ptr(B), _ptr(C_),
                                    float(self.kappaC_arr[0]), _ptr(yx), _ptr(iny), _ptr(inx), rho, 0, _ptr(T), _ptr(UC),
                                    _ptr(Sigma), _ptr(kappa), MEM_HOST))


LAKERNEL = {"Cholesky": HipCholKernel, "Eigen": HipEigenKernel, "Iterative": HipIterKernel, "Empirical": HipEmpirKernel}


def register(outstamp_cls, names=("Cholesky", "Eigen", "Iterative", "Empirical")):
    """Swap the HIP kernels into ``OutStamp.LAKERNEL`` (coadd.py:839-844) under the given keys."""
    for name in names:
        outstamp_cls.LAKERNEL[name] = LAKERNEL[name]
