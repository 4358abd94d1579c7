"""CPU oracle for the IMCOM postage-stamp path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing
under pyimcom_amd/ does.  It restates the reference's algorithm (file:line cited per function, paths
relative to the reference's src/pyimcom/) in NumPy/SciPy (LAPACK potrf/potrs, syevd, pocketfft) plus the
plain-C routines of oracle/imcom_oracle.c.

Parity status: PINNED -- tests/test_oracle.py checks every function here against golden vectors
produced by running the reference itself: routine.py / lakernel.py / psfutil.py imported by file path
(tests/golden/make_golden.py, make_golden_iter.py, make_golden_psf.py), and the stamp-driver functions of
coadd.py (make_selection, _process_input_stamps, trapezoid, _perform_coaddition, compress_map, smooth_and_pad,
the partition loop, _output_stamp_wrapper, the boundary recovery, the Iterative clamp of
_build_system_matrices) executed out of the module's syntax tree (make_golden_coadd.py, _smooth.py,
_partition.py, _block.py, _clamp.py); make_golden_chain.py runs one stamp through the whole unmodified chain.
"""

import ctypes as C
import os
import subprocess
from itertools import combinations

import numpy as np
from scipy.linalg import LinAlgError, cho_solve, cholesky

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    """Compile oracle/imcom_oracle.c with gcc (no FMA contraction)."""
    src = os.path.join(_HERE, "imcom_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def set_threads(n=0):
    """Threads of the C interpolators (0 = all host cores); LAPACK/BLAS threads are numpy's own (threadpoolctl)."""
    _c().orc_set_threads(int(n))


def _c():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _req(a, writable=False):
    assert isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous, "need C-contiguous float64"
    return a


# ------------------------------------------------------------------------------------------------ IN-1..4
def iD5512C_getw(w, fh):
    """routine.py:29-122."""
    _c().orc_d5512_getw(_p(_req(w)), C.c_double(fh))


def iD5512C(infunc, xpos, ypos, fhatout):
    """routine.py:125-181."""
    nlayer, ngy, ngx = infunc.shape
    _c().orc_interp_d5512(_p(_req(infunc)), nlayer, ngy, ngx, _p(_req(xpos)), _p(_req(ypos)), C.c_long(xpos.size),
                          _p(_req(fhatout)))


def iD5512C_sym(infunc, xpos, ypos, fhatout):
    """routine.py:184-253."""
    nlayer, ngy, ngx = infunc.shape
    _c().orc_interp_d5512_sym(_p(_req(infunc)), nlayer, ngy, ngx, _p(_req(xpos)), _p(_req(ypos)),
                              C.c_long(xpos.size), _p(_req(fhatout)))


def gridD5512C(infunc, xpos, ypos, fhatout):
    """routine.py:256-338."""
    ngy, ngx = infunc.shape
    npi, nxo = xpos.shape
    nyo = ypos.shape[1]
    _c().orc_grid_d5512(_p(_req(infunc)), ngy, ngx, _p(_req(xpos)), _p(_req(ypos)), C.c_long(npi), nxo, nyo,
                        _p(_req(fhatout)))


def lakernel1(lam, Q, mPhalf, C_, targetleak, kCmin, kCmax, nbis, kappa, Sigma, UC, T, smax):
    """routine.py:341-430 (Q unused, as in the reference)."""
    m, n = mPhalf.shape
    d = C.c_double
    _c().orc_lakernel1(_p(_req(lam)), _p(_req(mPhalf)), C.c_long(m), C.c_long(n), d(C_), d(targetleak), d(kCmin),
                       d(kCmax), int(nbis), _p(_req(kappa)), _p(_req(Sigma)), _p(_req(UC)), _p(_req(T)), d(smax))


def lsolve_sps(N, A, x, b):
    """routine.py:433-484 (A is destroyed)."""
    _c().orc_lsolve_sps(int(N), _p(_req(A)), _p(_req(x)), _p(_req(b)))


def build_reduced_T_wrap(Nflat, Dflat, Eflat, kappa, ucmin, smax, out_kappa, out_Sigma, out_UC, out_w):
    """routine.py:487-588."""
    d = C.c_double
    kappa = np.ascontiguousarray(kappa, dtype=np.float64)
    _c().orc_build_reduced_T(_p(_req(Nflat)), _p(_req(Dflat)), _p(_req(Eflat)), _p(kappa), int(kappa.size),
                             C.c_long(out_kappa.size), d(ucmin), d(smax), _p(_req(out_kappa)), _p(_req(out_Sigma)),
                             _p(_req(out_UC)), _p(_req(out_w)))


# ------------------------------------------------------------------------------------------------ CH / EI
def cholesky_wrapper(AA, A):
    """lakernel.py:241-279: lower Cholesky; on failure shift by |lambda_min(A)| + 1e-16, retry, restore.

    Returns (L, repaired)."""
    try:
        return cholesky(AA, lower=True, check_finite=False), False
    except LinAlgError:
        w = np.linalg.eigvalsh(A)
        di = np.diag_indices(A.shape[0])
        AA[di] += np.abs(w[0]) + 1e-16
        L = cholesky(AA, lower=True, check_finite=False)
        AA[di] -= np.abs(w[0]) + 1e-16
        return L, True


def chol_kernel(A, mBhalf, C_, kappaC, ucmin, smax, timings=None):
    """lakernel.CholKernel for ONE target PSF: A [n,n], mBhalf [m,n], C_ scalar.

    Single kappa: lakernel.py:281-323.  Multi kappa: 325-394.  Returns T f32 [m,n], UC, Sigma, kappa f32 [m],
    info (0, or node+1 of the first repaired factorisation).  ``timings`` (single kappa): dict that receives the
    seconds of the factorisation and of the triangular solves."""
    import time
    kappaC = np.atleast_1d(np.asarray(kappaC, dtype=np.float64))
    nv = kappaC.size
    m, n = mBhalf.shape
    T = np.zeros((m, n), dtype=np.float32)
    UC, Sigma, kappa = (np.zeros((m,), dtype=np.float32) for _ in range(3))
    info = 0
    if nv == 1:
        AA = A.flatten()
        my_kappa = kappaC[0] * C_
        if my_kappa:
            AA[:: n + 1] += my_kappa
        AA = AA.reshape((n, n))
        t0 = time.perf_counter()
        L, rep = cholesky_wrapper(AA, A)
        t1 = time.perf_counter()
        info = 1 if rep else 0
        Ti = cho_solve((L, True), mBhalf.T, check_finite=False).T
        if timings is not None:
            timings["factor"] = timings.get("factor", 0.0) + t1 - t0
            timings["tri_solve"] = timings.get("tri_solve", 0.0) + time.perf_counter() - t1
        D = np.einsum("ai,ai->a", mBhalf, Ti)
        N = np.einsum("ai,ai->a", Ti, Ti)
        kappa[:] = my_kappa
        Sigma[:] = N
        UC[:] = 1.0 - (my_kappa * N + D) / C_
        T[:, :] = Ti
        return T, UC, Sigma, kappa, info
    Tpi = np.zeros((nv, m, n))
    AA = np.copy(A)
    di = np.diag_indices(n)
    kappa_arr = kappaC * C_
    for j in range(nv):
        AA[di] += kappa_arr[j] - (kappa_arr[j - 1] if j > 0 else 0)
        L, rep = cholesky_wrapper(AA, A)
        if rep and info == 0:
            info = j + 1
        Tpi[j] = cho_solve((L, True), mBhalf.T, check_finite=False).T
    Dp = np.einsum("ai,pai->ap", mBhalf, Tpi)
    Npq = np.einsum("pai,qai->apq", Tpi, Tpi)
    Epq = np.zeros((m, nv, nv))
    for p in range(nv):
        for q in range(p):
            Epq[:, q, p] = Epq[:, p, q] = Dp[:, q] - kappa_arr[p] * Npq[:, p, q]
        Epq[:, p, p] = Dp[:, p] - kappa_arr[p] * Npq[:, p, p]
    ok, oS, oU, ow = np.zeros(m), np.zeros(m), np.zeros(m), np.zeros(m * nv)
    build_reduced_T_wrap(Npq.flatten(), Dp.flatten() / C_, Epq.flatten() / C_, kappaC, ucmin, smax, ok, oS, oU, ow)
    kappa[:] = ok * C_
    Sigma[:] = oS
    UC[:] = oU
    T[:, :] = np.einsum("pai,ap->ai", Tpi, ow.reshape((m, nv)))
    return T, UC, Sigma, kappa, info


def eigen_kernel(A, mBhalf, C_, kappaC, ucmin, smax, nbis=13):
    """lakernel.EigenKernel for ONE target PSF.  Single kappa 154-172; multi kappa 174-223 (incl. the
    kappa *= C of line 222 on top of kCmin*C, kCmax*C passed at 213-214)."""
    kappaC = np.atleast_1d(np.asarray(kappaC, dtype=np.float64))
    m, n = mBhalf.shape
    T = np.zeros((m, n), dtype=np.float32)
    UC, Sigma, kappa = (np.zeros((m,), dtype=np.float32) for _ in range(3))
    lam, Q = np.linalg.eigh(A)
    mPhalf = mBhalf @ Q
    if kappaC.size == 1:
        k = kappaC[0] * C_
        kappa[:] = k
        Sigma[:] = np.sum((mPhalf / (lam + k)) ** 2, axis=1)
        UC[:] = 1 - (lam + 2 * k) / (lam + k) ** 2 @ mPhalf.T**2 / C_
        T[:, :] = mPhalf / (lam + k) @ Q.T
        return T, UC, Sigma, kappa, 0
    tt = np.zeros((m, n))
    k64, S64, U64 = np.zeros(m), np.zeros(m), np.zeros(m)
    lakernel1(lam, Q, np.ascontiguousarray(mPhalf), C_, ucmin, kappaC[0] * C_, kappaC[-1] * C_, nbis, k64, S64, U64,
              tt, smax)
    kappa[:] = k64  # float32 store, then *= C in float32 as lakernel.py:216-222 does
    Sigma[:] = S64
    UC[:] = U64
    kappa *= C_
    T[:, :] = tt @ Q.T
    return T, UC, Sigma, kappa, 0


def conjugate_gradient(A, b, rtol=1.5e-3, maxiter=30, info=None):
    """lakernel.conjugate_gradient (lakernel.py:397-442): plain CG from x = 0, stop when |r| < rtol |b|.  ``info`` (a list, test
    instrumentation the reference does not have): receives the number of steps taken."""
    atol = np.linalg.norm(b) * rtol
    x = np.zeros_like(b)
    r = b.copy()
    rho_prev = 0.0
    p = r.copy()
    steps = 0
    for iteration in range(maxiter):
        rho_cur = np.dot(r, r)
        if rho_cur**0.5 < atol:
            break
        if iteration > 0:
            p *= rho_cur / rho_prev
            p += r
        q = A @ p
        alpha = rho_cur / np.dot(p, q)
        x += alpha * p
        r -= alpha * q
        rho_prev = rho_cur
        steps += 1
    if info is not None:
        info.append(steps)
    return x


def _relevant(out_y, out_x, in_y, in_x, rho_acc):
    """Acceptance mask of lakernel.py:617-622: input pixel within rho_acc of the output pixel."""
    return np.hypot(out_y[:, None] - in_y[None, :], out_x[:, None] - in_x[None, :]) < rho_acc


def _iterative_wrapper(AA, mBhalf, relevant, rtol, maxiter, steps=None):
    """lakernel.IterKernel._iterative_wrapper (545-586): one restricted CG solve per output pixel, float32 T.  ``steps`` (a list,
    test instrumentation): receives the CG steps of every output pixel."""
    m, n = mBhalf.shape
    Ti = np.zeros((m, n), dtype=np.float32)
    for a in range(m):
        sel = np.nonzero(relevant[a])[0]
        Ti[a, sel] = conjugate_gradient(AA[np.ix_(sel, sel)], mBhalf[a, sel], rtol, maxiter, info=steps)
    return Ti


def iter_kernel(A, mBhalf, C_, kappaC, ucmin, smax, out_y, out_x, in_y, in_x, rho_acc, rtol=1.5e-3, maxiter=30,
                exact_UC=None, steps=None):
    """lakernel.IterKernel for ONE target PSF (single kappa 588-654, multi kappa 656-744).  exact_UC None takes the
    reference defaults (False for one node, True for several)."""
    kappaC = np.atleast_1d(np.asarray(kappaC, dtype=np.float64))
    nv = kappaC.size
    m, n = mBhalf.shape
    T = np.zeros((m, n), dtype=np.float32)
    UC, Sigma, kappa = (np.zeros((m,), dtype=np.float32) for _ in range(3))
    relevant = _relevant(out_y, out_x, in_y, in_x, rho_acc)
    di = np.diag_indices(n)
    if nv == 1:
        exact = False if exact_UC is None else exact_UC
        AA = np.copy(A)
        my_kappa = kappaC[0] * C_
        if my_kappa:
            AA[di] += my_kappa
        Ti = _iterative_wrapper(AA, mBhalf, relevant, rtol, maxiter, steps=steps)
        D = np.einsum("ai,ai->a", mBhalf, Ti)
        N = np.einsum("ai,ai->a", Ti, Ti)
        kappa[:] = my_kappa
        Sigma[:] = N
        if exact:
            E = np.einsum("ij,ai,aj->a", A, Ti, Ti)
            UC[:] = 1.0 + (E - 2 * D) / C_
        else:
            UC[:] = 1.0 - (my_kappa * N + D) / C_
        T[:, :] = Ti
        return T, UC, Sigma, kappa, 0
    exact = True if exact_UC is None else exact_UC
    Tpi = np.zeros((nv, m, n))
    AA = np.copy(A)
    kappa_arr = kappaC * C_
    for j in range(nv):
        AA[di] += kappa_arr[j] - (kappa_arr[j - 1] if j > 0 else 0)
        Tpi[j] = _iterative_wrapper(AA, mBhalf, relevant, rtol, maxiter)
    Dp = np.einsum("ai,pai->ap", mBhalf, Tpi)
    Npq = np.einsum("pai,qai->apq", Tpi, Tpi)
    Epq = np.zeros((m, nv, nv))
    for p in range(nv):
        for q in range(p + 1):
            if exact:
                Epq[:, q, p] = Epq[:, p, q] = np.einsum("ij,ai,aj->a", A, Tpi[p], Tpi[q])
            else:
                Epq[:, q, p] = Epq[:, p, q] = Dp[:, q] - kappa_arr[p] * Npq[:, p, q]
    ok, oS, oU, ow = np.zeros(m), np.zeros(m), np.zeros(m), np.zeros(m * nv)
    build_reduced_T_wrap(Npq.flatten(), Dp.flatten() / C_, Epq.flatten() / C_, kappaC, ucmin, smax, ok, oS, oU, ow)
    kappa[:] = ok * C_
    Sigma[:] = oS
    UC[:] = oU
    T[:, :] = np.einsum("pai,ap->ai", Tpi, ow.reshape((m, nv)))
    return T, UC, Sigma, kappa, 0


def empir_kernel(A, mBhalf, C_, kappaC, out_y, out_x, in_y, in_x, rho_acc, no_qlt_ctrl=False):
    """lakernel.EmpirKernel for ONE target PSF (747-805): T_ai = max(rho - dist, 0) / row sum, no linear solve."""
    kappaC = np.atleast_1d(np.asarray(kappaC, dtype=np.float64))
    m, n = mBhalf.shape
    Ti = np.maximum(rho_acc - np.hypot(out_y[:, None] - in_y[None, :], out_x[:, None] - in_x[None, :]), 0)
    with np.errstate(invalid="ignore", divide="ignore"):
        Ti /= np.sum(Ti, axis=-1)[:, None]
    T = Ti.astype(np.float32)
    UC, Sigma, kappa = (np.zeros((m,), dtype=np.float32) for _ in range(3))
    if no_qlt_ctrl:
        return T, UC, Sigma, kappa, 0
    D = np.einsum("ai,ai->a", mBhalf, Ti)
    N = np.einsum("ai,ai->a", Ti, Ti)
    E = np.einsum("ij,ai,aj->a", A, Ti, Ti)
    kappa[:] = kappaC[0] * C_
    Sigma[:] = N
    UC[:] = 1.0 + (E - 2 * D) / C_
    return T, UC, Sigma, kappa, 0


def la_kernel(kind, A, mhalfb, outovlc, n2f, kappaC, ucmin, smax):
    """_LAKernel.__call__ (lakernel.py:84-138) over n_out target PSFs, incl. the n == 0 case (110-119)."""
    n_out, m, n = mhalfb.shape
    shape = (n_out, n2f, n2f)
    if n == 0:
        return (np.zeros((n_out, m, 0), np.float32), np.ones(shape, np.float32), np.zeros(shape, np.float32),
                np.ones(shape, np.float32))
    fn = {"Cholesky": chol_kernel, "Eigen": eigen_kernel}[kind]
    T = np.zeros((n_out, m, n), np.float32)
    UC, Sigma, kappa = (np.zeros((n_out, m), np.float32) for _ in range(3))
    for k in range(n_out):
        T[k], UC[k], Sigma[k], kappa[k], _ = fn(A, mhalfb[k], outovlc[k], kappaC, ucmin, smax)
    return T, UC.reshape(shape), Sigma.reshape(shape), kappa.reshape(shape)


# ------------------------------------------------------------------------------------------------ PS-1..4
class Geom:
    """PSFGrp.setup / PSFOvl.setup constants (psfutil.py:568-613, 1065-1089), psfsplit off."""

    def __init__(self, npixpsf=48, oversamp=8, dtheta_deg=0.025 / 3600, flat_penalty=1e-7):
        self.oversamp = oversamp
        self.nsamp = npixpsf * oversamp - 1
        self.nc = self.nsamp // 2
        self.nfft = npixpsf * oversamp * 2
        arcsec = np.pi / 180.0 / 60.0 / 60.0  # config.py:85-98 (Settings.arcsec, pixscale_native = 0.11 arcsec)
        self.dscale = ((0.11 * arcsec) / arcsec) / oversamp / (dtheta_deg * 3600)  # psfutil.py:610
        self.flat_penalty = flat_penalty
        self.yxo = np.mgrid[(1 - self.nsamp) / 2 : (self.nsamp - 1) / 2 : self.nsamp * 1j,
                            (1 - self.nsamp) / 2 : (self.nsamp - 1) / 2 : self.nsamp * 1j]


def pad_and_rfft2(psf_arr, g):
    """PSFGrp.accel_pad_and_rfft2 (psfutil.py:943-986)."""
    n_arr = psf_arr.shape[0]
    pad_m1 = np.zeros((n_arr, g.nsamp, g.nfft))
    pad_m2 = np.zeros((n_arr, g.nfft, g.nfft // 2 + 1), dtype=np.complex128)
    pad_m1[:, :, : g.nsamp] = psf_arr
    pad_m2[:, : g.nsamp, :] = np.fft.rfft(pad_m1, axis=-1)
    return np.fft.fft(pad_m2, axis=-2)


def irfft2_and_extract(ovl_rft, g):
    """PSFOvl.accel_irfft2_and_extract, live branch (psfutil.py:1226-1227)."""
    nc = g.nc
    return np.roll(np.fft.irfft2(ovl_rft), nc, axis=(-2, -1))[..., : 2 * nc + 1, : 2 * nc + 1]


def overlap_cross(rft1, rft2, g):
    """PSFOvl._build_psfovl cross branch (psfutil.py:1259-1265): [n1, n2, nsamp, nsamp]."""
    out = np.zeros((rft1.shape[0], rft2.shape[0], g.nsamp, g.nsamp))
    for i in range(rft1.shape[0]):
        out[i] = irfft2_and_extract(rft1[i] * rft2.conjugate(), g)
    return out


def tri_index(n_psf, i1, i2):
    """PSFOvl._idx_square2triangle (psfutil.py:1139-1175)."""
    assert i1 <= i2
    return (2 * n_psf - i1 + 1) * i1 // 2 + i2 - i1


def overlap_self(rft, g):
    """input self-overlap branch (psfutil.py:1270-1278): triangle storage [n(n+1)/2, nsamp, nsamp]."""
    n_psf = rft.shape[0]
    out = np.zeros((n_psf * (n_psf + 1) // 2, g.nsamp, g.nsamp))
    for i in range(n_psf):
        start = tri_index(n_psf, i, i)
        out[start : start + n_psf - i] = irfft2_and_extract(rft[i] * rft[i:].conjugate(), g)
    return out


def overlap_out_C(rft_out, g):
    """output self-overlap -> C (psfutil.py:1283-1290)."""
    ovl = irfft2_and_extract(rft_out * rft_out.conjugate(), g)
    return ovl[:, g.nc, g.nc]


def psf_gaussian(n, sigmax, sigmay):
    """OutPSF.psf_gaussian (psfutil.py:117-146)."""
    y, x = np.mgrid[(1 - n) / 2 / sigmay : (n - 1) / 2 / sigmay : n * 1j, (1 - n) / 2 / sigmax : (n - 1) / 2 / sigmax : n * 1j]
    return np.exp(-0.5 * (np.square(x) + np.square(y))) / (2.0 * np.pi * sigmax * sigmay)


def psf_simple_airy(n, ldp, obsc=0.0, tophat_conv=0.0, sigma=0.0):
    """OutPSF.psf_simple_airy (psfutil.py:148-223)."""
    from scipy.special import jv

    kp = 1 + int(np.ceil(tophat_conv + 6 * sigma))
    npad = n + 2 * kp
    y, x = np.mgrid[(1 - npad) / 2 : (npad - 1) / 2 : npad * 1j, (1 - npad) / 2 : (npad - 1) / 2 : npad * 1j]
    r = np.sqrt(np.square(x) + np.square(y)) / ldp
    I_ = (np.square(jv(0, np.pi * r) + jv(2, np.pi * r) - obsc**2 * (jv(0, np.pi * r * obsc) + jv(2, np.pi * r * obsc)))
          / (4.0 * ldp**2 * (1 - obsc**2)) * np.pi)
    It = np.fft.rfft2(I_)
    uxa = np.linspace(0, 1 - 1 / npad, npad)
    uxa[-(npad // 2):] -= 1
    ux = np.tile(uxa[None, : npad // 2 + 1], (npad, 1))
    uy = np.tile(uxa[:, None], (1, npad // 2 + 1))
    It *= (np.exp(-2.0 * np.pi**2 * (np.square(ux * sigma) + np.square(uy * sigma))) * np.sinc(ux * tophat_conv)
           * np.sinc(uy * tophat_conv))
    I_ = np.fft.irfft2(It, s=(npad, npad))
    return I_[kp:-kp, kp:-kp]


# ------------------------------------------------------------------------------------------------ PS-5/6
def subblock_ii_self(ovl_tri, n_psf, g, x1, y1, cnt1, x2=None, y2=None, cnt2=None):
    """PSFOvl._call_ii_self (psfutil.py:1597-1732).  cnt = pixels per exposure (pix_count); exposures index
    the group's PSFs directly (idx_blk2grp = identity)."""
    same = x2 is None
    if same:
        x2, y2, cnt2 = x1, y1, cnt1
    cs1 = np.concatenate([[0], np.cumsum(cnt1)]).astype(int)
    cs2 = np.concatenate([[0], np.cumsum(cnt2)]).astype(int)
    res = np.zeros((cs1[-1], cs2[-1]))
    ddx = x1[:, None] - x2[None, :]
    ddy = y1[:, None] - y2[None, :]
    ddx /= g.dscale
    ddx += g.nc
    ddy /= g.dscale
    ddy += g.nc
    for j in range(len(cnt1)):
        if cnt1[j] == 0:
            continue
        for i in range(0 if not same else j, len(cnt2)):
            if cnt2[i] == 0:
                continue
            tab = ovl_tri[tri_index(n_psf, j, i)] if j <= i else np.flip(ovl_tri[tri_index(n_psf, i, j)])
            sl = np.s_[cs1[j] : cs1[j + 1], cs2[i] : cs2[i + 1]]
            out = np.zeros((1, int(cnt1[j]) * int(cnt2[i])))
            fn = iD5512C_sym if (same and j == i) else iD5512C
            fn(np.ascontiguousarray(np.pad(tab, 6)).reshape((1, g.nsamp + 12, g.nsamp + 12)),
               np.ascontiguousarray(ddx[sl].ravel() + 6), np.ascontiguousarray(ddy[sl].ravel() + 6), out)
            res[sl] = out.reshape((int(cnt1[j]), int(cnt2[i])))
            if g.flat_penalty != 0.0:
                res[sl] -= g.flat_penalty / n_psf
                if j == i:
                    res[sl] += g.flat_penalty
            if same and j < i:
                res[cs2[i] : cs2[i + 1], cs1[j] : cs1[j + 1]] = res[sl].T
    return res


def subblock_ii_cross(ovl, g, x1, y1, cnt1, x2, y2, cnt2, ids1=None, ids2=None):
    """PSFOvl._call_ii_cross (psfutil.py:1401-1495); ovl [n1, n2, nsamp, nsamp].  cnt1 / cnt2: pixels per PSF of
    each group; ids1 / ids2: the block exposure index of each of those PSFs (idx_grp2blk; default: identical
    numbering), which is what the same-exposure term of the flat penalty compares (lines 1483-1486)."""
    ids1 = np.arange(len(cnt1)) if ids1 is None else np.asarray(ids1)
    ids2 = np.arange(len(cnt2)) if ids2 is None else np.asarray(ids2)
    cs1 = np.concatenate([[0], np.cumsum(cnt1)]).astype(int)
    cs2 = np.concatenate([[0], np.cumsum(cnt2)]).astype(int)
    res = np.zeros((cs1[-1], cs2[-1]))
    ddx = x1[:, None] - x2[None, :]
    ddx /= g.dscale
    ddx += g.nc
    ddy = y1[:, None] - y2[None, :]
    ddy /= g.dscale
    ddy += g.nc
    n_in = (ovl.shape[0] * ovl.shape[1]) ** 0.5
    for j in range(len(cnt1)):
        if cnt1[j] == 0:
            continue
        for i in range(len(cnt2)):
            if cnt2[i] == 0:
                continue
            sl = np.s_[cs1[j] : cs1[j + 1], cs2[i] : cs2[i + 1]]
            out = np.zeros((1, int(cnt1[j]) * int(cnt2[i])))
            iD5512C(np.ascontiguousarray(np.pad(ovl[j, i], 6)).reshape((1, g.nsamp + 12, g.nsamp + 12)),
                    np.ascontiguousarray(ddx[sl].ravel() + 6), np.ascontiguousarray(ddy[sl].ravel() + 6), out)
            res[sl] = out.reshape((int(cnt1[j]), int(cnt2[i])))
            if g.flat_penalty != 0.0:
                res[sl] -= g.flat_penalty / n_in
                if ids1[j] == ids2[i]:
                    res[sl] += g.flat_penalty
    return res


def subblock_io(ovl_io, g, x_in, y_in, cnt, out_x, out_y, selection=None):
    """PSFOvl._call_io_cross (psfutil.py:1497-1595); ovl_io [n_in, n_out, nsamp, nsamp];
    out_x [nxo], out_y [nyo] output pixel coordinates -> res [n_out, nyo*nxo, n_sel]."""
    cs = np.concatenate([[0], np.cumsum(cnt)]).astype(int)
    if selection is not None:
        x_in, y_in = x_in[selection], y_in[selection]
        cs = np.searchsorted(selection, cs)
    cnt_ = np.diff(cs)
    n_outpix = out_x.size * out_y.size
    res = np.zeros((ovl_io.shape[1], n_outpix, x_in.size))
    ddx = x_in[:, None] - out_x[None, :]
    ddx /= g.dscale
    ddx += g.nc
    ddy = y_in[:, None] - out_y[None, :]
    ddy /= g.dscale
    ddy += g.nc
    for k in range(ovl_io.shape[1]):
        for j in range(len(cnt)):
            if cnt[j] == 0:
                continue
            out = np.zeros((int(cnt_[j]), n_outpix))
            gridD5512C(np.ascontiguousarray(np.pad(ovl_io[j, k], 6)), np.ascontiguousarray(ddx[cs[j] : cs[j + 1], :] + 6),
                       np.ascontiguousarray(ddy[cs[j] : cs[j + 1], :] + 6), out)
            res[k, :, cs[j] : cs[j + 1]] = out.T
    return res


def stamp_system_groups(instamps, selections, groups, rft_in, rft_out, g, out_x, out_y, group_expo=None):
    """OutStamp._build_system_matrices (coadd.py:1027-1082) with the sub-blocks SysMatA / SysMatB hand out
    (psfutil.py:1904-2010, 2128-2199): nine InStamps (x_val, y_val, data, pix_cumsum) or None, their selections
    (index arrays or None = all), the key of the 2x2 PSF group each belongs to, rft_in[key] = that group's PSF
    transforms.  Sub-blocks between InStamps of one group come from the group's self overlap (_call_ii_self), between
    groups from PSFOvl(group of the first InStamp, group of the second) (_call_ii_cross).  group_expo[key] lists the
    block exposures the group holds PSFs for (idx_grp2blk, psfutil.py:820-832; default all): InStamps of the group
    have no pixels from other exposures.  Returns A [N, N] and -B/2 [m, N] for the first target PSF."""
    present = [k for k in range(9) if instamps[k] is not None]
    sels = {k: (np.arange(instamps[k][0].size) if selections[k] is None else np.asarray(selections[k])) for k in present}
    counts = [sels[k].size if k in sels else 0 for k in range(9)]
    cum = np.cumsum([0] + counts)
    A = np.zeros((cum[-1], cum[-1]))
    tri, cross = {}, {}
    ge = (lambda k: np.arange(rft_in[k].shape[0])) if group_expo is None else (lambda k: np.asarray(group_expo[k]))
    cnt = {k: np.diff(instamps[k][3])[ge(groups[k])] for k in present}  # per PSF of the InStamp's group
    for k in present:
        assert cnt[k].sum() == instamps[k][0].size, "an InStamp has pixels from an exposure its PSF group lacks"
    for ia, a in enumerate(present):
        xa, ya = instamps[a][0], instamps[a][1]
        ga = groups[a]
        if ga not in tri:
            tri[ga] = overlap_self(rft_in[ga], g)
        sub = subblock_ii_self(tri[ga], rft_in[ga].shape[0], g, xa, ya, cnt[a])
        A[cum[a] : cum[a + 1], cum[a] : cum[a + 1]] = sub[np.ix_(sels[a], sels[a])]
        for b in present[ia + 1 :]:
            xb, yb = instamps[b][0], instamps[b][1]
            gb = groups[b]
            if ga == gb:
                sub = subblock_ii_self(tri[ga], rft_in[ga].shape[0], g, xa, ya, cnt[a], xb, yb, cnt[b])
            else:
                if (ga, gb) not in cross:
                    cross[(ga, gb)] = overlap_cross(rft_in[ga], rft_in[gb], g)
                sub = subblock_ii_cross(cross[(ga, gb)], g, xa, ya, cnt[a], xb, yb, cnt[b], ge(ga), ge(gb))
            sub = sub[np.ix_(sels[a], sels[b])]
            A[cum[a] : cum[a + 1], cum[b] : cum[b + 1]] = sub
            A[cum[b] : cum[b + 1], cum[a] : cum[a + 1]] = sub.T
    mB = np.zeros((out_x.size * out_y.size, cum[-1]))
    io = {}
    for a in present:
        ga = groups[a]
        if ga not in io:
            io[ga] = overlap_cross(rft_in[ga], rft_out, g)
        sel = None if selections[a] is None else sels[a]
        mB[:, cum[a] : cum[a + 1]] = subblock_io(io[ga], g, instamps[a][0], instamps[a][1], cnt[a], out_x, out_y, sel)[0]
    return A, mB


# ------------------------------------------------------------------------------------------------ ST-1..4
def trapezoid(arr, fade_kernel):
    """OutStamp.trapezoid, default arguments (coadd.py:1222-1282), in place on (..., ny, nx)."""
    fk2 = fade_kernel * 2
    if not fk2 > 0:
        return
    ny, nx = arr.shape[-2:]
    it, ir = ny - 1, nx - 1
    s = np.arange(1, fk2 + 1, dtype=np.float64) / (fk2 + 1)
    s -= np.sin(2 * np.pi * s) / (2 * np.pi)
    sT = s[None, :].T
    arr[..., 0:fk2, :] *= sT
    arr[..., it : it - fk2 : -1, :] *= sT
    arr[..., :, 0:fk2] *= s
    arr[..., :, ir : ir - fk2 : -1] *= s


def iterative_clamp(UC, Sigma):
    """OutStamp._build_system_matrices after the "Iterative" kernel (coadd.py:1104-1107): UC and Sigma "could be
    negative as the iterative kernel is not exact" and are raised to 1e-32.  Returns new arrays (np.maximum)."""
    return np.maximum(UC, 1e-32), np.maximum(Sigma, 1e-32)


def trapezoid_recover(arr, fade_kernel, pad_widths=(0, 0, 0, 0)):
    """OutStamp.trapezoid(..., recover_mode=True, pad_widths) (coadd.py:1262-1292), in place."""
    fk2 = fade_kernel * 2
    if not fk2 > 0:
        return
    ny, nx = arr.shape[-2:]
    pb, pt, pl, pr = pad_widths
    it, ir = ny - pt - 1, nx - pr - 1
    s = np.arange(1, fk2 + 1, dtype=np.float64) / (fk2 + 1)
    s -= np.sin(2 * np.pi * s) / (2 * np.pi)
    sT = s[None, :].T
    arr[..., pb : pb + fk2, :] /= sT
    arr[..., it : it - fk2 : -1, :] /= sT
    arr[..., :, pl : pl + fk2] /= s
    arr[..., :, ir : ir - fk2 : -1] /= s


def block_accumulate(dst, src, j_st, i_st, n2, fade_kernel):
    """One map update of Block._output_stamp_wrapper (coadd.py:1975-1993): dst[..., bottom:top, left:right] += src."""
    bottom, left = (j_st - 1) * n2, (i_st - 1) * n2
    top, right = j_st * n2 + fade_kernel * 2, i_st * n2 + fade_kernel * 2
    dst[..., bottom:top, left:right] += src


def stamp_loop_order(j_st_min, j_st_max, i_st_min, i_st_max, nrun=None):
    """The stamps Block.coadd_output_stamps visits, in its order (coadd.py:2056-2064): cells of 2 x 2 stamps, row by row of
    cells; inside a cell product(range(2), range(2)) = dj outer, di inner; the loop returns after nrun stamps."""
    from itertools import product

    out = []
    for j_st in range(j_st_min, j_st_max + 1, 2):
        for i_st in range(i_st_min, i_st_max + 1, 2):
            for dj, di in product(range(2), range(2)):
                out.append((j_st + dj, i_st + di))
                if len(out) == nrun:
                    return out
    return out


def compress_map(map_, coef, dtype):
    """Block.compress_map (coadd.py:2087-2138) without the FITS wrapping."""
    a_min, a_max = (0, 65535) if dtype == np.uint16 else (-32768, 32767)
    return np.clip(np.floor(coef * np.log10(np.clip(map_, 1e-32, None)) + 0.5), a_min, a_max).astype(dtype)


def perform_coaddition(T, indata, expo, n_expo, n2f, n2, fade_kernel, inpix_cumsum=None):
    """OutStamp._perform_coaddition (coadd.py:1294-1354) for one stamp.

    T f32 [n_out, m, N] (tapered in place), indata f32 [n_inframe, N], expo[N] = exposure of each pixel.  The
    reference walks the nine InStamps and adds, per InStamp and exposure, the float32 sum of that segment of T into a
    float64 accumulator (1329-1337): the segments are the runs of equal exposure inside each InStamp, delimited by
    inpix_cumsum [10] (without it: the maximal runs of equal exposure, the same thing unless two neighbouring
    InStamps meet on one exposure)."""
    n_out, m, N = T.shape
    if fade_kernel > 0:
        T_view = np.moveaxis(T, 1, -1).reshape((n_out, N, n2f, n2f))
        trapezoid(T_view, fade_kernel)
    Tsum_image = np.zeros((n_out, m, n_expo))
    expo = np.asarray(expo)
    cuts = np.flatnonzero(np.diff(expo)) + 1
    if inpix_cumsum is not None:
        cuts = np.union1d(cuts, np.asarray(inpix_cumsum, dtype=np.int64)[1:-1])
    bounds = np.concatenate([[0], cuts[(cuts > 0) & (cuts < N)], [N]]).astype(np.int64)
    for a, b in zip(bounds[:-1], bounds[1:]):
        if b > a:
            Tsum_image[:, :, int(expo[a])] += np.sum(T[:, :, a:b], axis=2)
    Tsum_stamp = np.sum(Tsum_image, axis=1) / n2**2
    Tsum_inpix = np.sum(Tsum_image, axis=2).reshape((n_out, n2f, n2f))
    Tsum_norm = Tsum_image / np.abs(Tsum_image).sum(axis=2)[:, :, None]
    Neff = 1.0 / np.sum(np.square(Tsum_norm), axis=2).reshape((n_out, n2f, n2f))
    if fade_kernel > 0:
        trapezoid(Neff, fade_kernel)
    outimage = np.einsum("oaj,ij->oia", T, indata).reshape((n_out, indata.shape[0], n2f, n2f))
    return outimage, Tsum_stamp, Tsum_inpix, Neff


def select_pixels(x, y, pivot, radius):
    """InStamp.make_selection (coadd.py:716-749): indices with dist^2 < radius^2 from the pivot, or None."""
    if pivot == (None, None) or radius is None:
        return None
    dist_sq = np.zeros(x.shape)
    if pivot[0] is not None:
        dist_sq += np.square(x - pivot[0])
    if pivot[1] is not None:
        dist_sq += np.square(y - pivot[1])
    sel = np.where(dist_sq < radius**2)[0]
    return sel if sel.shape[0] < x.shape[0] else None


def partition_pixels(out_x, out_y, in_x, in_y, mask, use_instamps, n2, n1P, npixmax):
    """The binning loop of InImage.partition_pixels (coadd.py:329-358), pixels given in visiting order."""
    nst = n1P + 2
    pix_lower, pix_upper = -n2 - 0.5, n1P * n2 + n2 - 0.5
    y_idx = np.zeros((nst, nst, npixmax), dtype=np.uint16)
    x_idx = np.zeros((nst, nst, npixmax), dtype=np.uint16)
    y_val = np.zeros((nst, nst, npixmax))
    x_val = np.zeros((nst, nst, npixmax))
    pix_count = np.zeros((nst, nst), dtype=np.uint32)
    for p in range(len(out_x)):
        my_x, my_y = out_x[p], out_y[p]
        if not (pix_lower < my_x < pix_upper and pix_lower < my_y < pix_upper):
            continue
        if mask is not None and not mask[p]:
            continue
        i_st = int((my_x - pix_lower) // n2)
        j_st = int((my_y - pix_lower) // n2)
        if not use_instamps[j_st, i_st]:
            continue
        k = pix_count[j_st, i_st]
        y_idx[j_st, i_st, k], x_idx[j_st, i_st, k] = in_y[p], in_x[p]
        y_val[j_st, i_st, k], x_val[j_st, i_st, k] = my_y, my_x
        pix_count[j_st, i_st] += 1
    return y_idx, x_idx, y_val, x_val, pix_count


def process_input_stamps(instamps, pivots, radius):
    """OutStamp._process_input_stamps (coadd.py:886-977): concatenate the selections of the nine neighbours.

    instamps: nine entries (x_val, y_val, data [n_inframe, k], pix_cumsum) or None; pivots: nine (x_pivot, y_pivot)
    with None entries.  Returns inx_val, iny_val, indata, exposure index per pixel, inpix_cumsum [10]."""
    xs, ys, ds, es, counts = [], [], [], [], []
    for inst, pivot in zip(instamps, pivots):
        if inst is None:
            counts.append(0)
            continue
        x, y, data, cum = inst
        expo = np.repeat(np.arange(len(cum) - 1), np.diff(cum))
        sel = select_pixels(x, y, tuple(pivot), radius)
        if sel is None:
            sel = np.arange(x.shape[0])
        xs.append(x[sel]); ys.append(y[sel]); ds.append(data[:, sel]); es.append(expo[sel])
        counts.append(sel.shape[0])
    n_inframe = next(i[2].shape[0] for i in instamps if i is not None)
    cat = lambda parts, empty: np.hstack(parts) if parts else empty  # noqa: E731
    return (cat(xs, np.zeros(0)), cat(ys, np.zeros(0)), cat(ds, np.zeros((n_inframe, 0), np.float32)),
            cat(es, np.zeros(0, int)), np.cumsum([0] + counts))


QFILTER_NATIVE = [1.155, 1.456, 1.250, 1.021, 0.834, 0.689, 0.491, 1.009, 0.000, 1.159, 1.685]  # config.py:85-98
OBSC = 0.31


def sample_psf(psf, nsamp, yxco=None):
    """PSFGrp._sample_psf (psfutil.py:709-795) for ONE image psf [ny, nx]: interpolate at the nsamp x nsamp sampling
    positions (yxco [2, nsamp, nsamp] = y, x offsets from the image centre; None = the unrotated grid, which the
    reference evaluates with the separable gridD5512C instead of iD5512C)."""
    ny, nx = psf.shape
    xctr, yctr = (nx - 1) / 2.0, (ny - 1) / 2.0
    out_arr = np.zeros((1, nsamp * nsamp))
    lin = np.linspace((1 - nsamp) / 2, (nsamp - 1) / 2, nsamp)
    if yxco is not None:
        iD5512C(np.pad(psf, 6).reshape((1, ny + 12, nx + 12)), np.ascontiguousarray(yxco[1].ravel() + xctr + 6),
                np.ascontiguousarray(yxco[0].ravel() + yctr + 6), out_arr)
    else:
        gridD5512C(np.pad(psf, 6), (lin + xctr + 6)[None, :], (lin + yctr + 6)[None, :], out_arr)
    return out_arr.reshape((nsamp, nsamp))


def finish_psf_group(psf_arr, psf_circ, psf_norm):
    """PSFGrp.__init__ (psfutil.py:650-656): circular cut-out and normalisation of the sampled PSFs (in place)."""
    nsamp = psf_arr.shape[-1]
    lin = np.linspace((1 - nsamp) / 2, (nsamp - 1) / 2, nsamp)
    if psf_circ:
        psf_arr *= np.hypot(lin[:, None], lin[None, :]) < nsamp // 2 + 0.5
    if psf_norm:
        v = np.moveaxis(psf_arr, 0, -1)
        v /= psf_arr.sum(axis=(-2, -1))
    return psf_arr


def get_outpsf(outpsf, extrasmooth, use_filter, nsamp, oversamp):
    """PSFGrp._get_outpsf (psfutil.py:854-896): the (nsamp+1)^2 target PSF image before sampling."""
    if outpsf == "GAUSSIAN":
        return psf_gaussian(nsamp + 1, extrasmooth * oversamp, extrasmooth * oversamp)
    if outpsf in ("AIRYOBSC", "AIRYUNOBSC"):
        return psf_simple_airy(nsamp + 1, QFILTER_NATIVE[use_filter] * oversamp, obsc=OBSC if outpsf == "AIRYOBSC" else 0.0,
                               tophat_conv=0.0, sigma=extrasmooth * oversamp)
    raise RuntimeError("Error: unsupported target output PSF type")


# ------------------------------------------------------------------------------------------------ stamp level
def stamp_system(g, x, y, psf, tables_pad, pair_tab, pair_pen, io_tab, out_x0, out_y0, n2f):
    """A and Bt of ONE stamp from the device-seam description (include/imcom_hip.h: imcom_build_A/_B):
    element (i,j), i <= j, is the D5512 interpolation of the pair's table at ((x_i-x_j)/dscale+nc+6, ...),
    mirrored -- which is what the pinned sub-block functions above produce block by block (the A assembly
    of coadd.py:1027-1068 only copies and transposes sub-blocks)."""
    from_pad = tables_pad
    n = x.size
    ng = from_pad.shape[-1]
    A = np.zeros((n, n))
    iu, ju = np.triu_indices(n)
    code = pair_tab[psf[iu], psf[ju]]
    pen = pair_pen[psf[iu], psf[ju]]
    vals = np.zeros(iu.size)
    for c in np.unique(code):
        if c < 0:
            continue
        sel = np.where(code == c)[0]
        swap, flip, tab = bool(c & (1 << 29)), bool(c & (1 << 30)), int(c & ((1 << 28) - 1))
        a, b = (ju[sel], iu[sel]) if swap else (iu[sel], ju[sel])
        ddx = x[a] - x[b]
        ddx /= g.dscale
        ddx += g.nc
        ddy = y[a] - y[b]
        ddy /= g.dscale
        ddy += g.nc
        t = from_pad[tab]
        if flip:
            t = np.ascontiguousarray(np.flip(t))
        out = np.zeros((1, sel.size))
        iD5512C(t.reshape((1, ng, ng)), np.ascontiguousarray(ddx + 6), np.ascontiguousarray(ddy + 6), out)
        vals[sel] = out[0]
    vals += pen
    A[iu, ju] = vals
    A[ju, iu] = vals
    m = n2f * n2f
    Bt = np.zeros((n, m))
    ox = out_x0 + np.arange(n2f, dtype=np.float64)
    oy = out_y0 + np.arange(n2f, dtype=np.float64)
    for p in np.unique(psf):
        sel = np.where(psf == p)[0]
        ddx = x[sel, None] - ox[None, :]
        ddx /= g.dscale
        ddx += g.nc
        ddy = y[sel, None] - oy[None, :]
        ddy /= g.dscale
        ddy += g.nc
        out = np.zeros((sel.size, m))
        gridD5512C(np.ascontiguousarray(from_pad[io_tab[p]]), np.ascontiguousarray(ddx + 6), np.ascontiguousarray(ddy + 6), out)
        Bt[sel] = out
    return A, Bt


def smooth_and_pad(image, tophatwidth=0.0, gaussiansigma=0.0):
    """InImage.smooth_and_pad (coadd.py:433-474): pad by npad = ceil(w + 6 s + 1) rounded up to a multiple of 4, multiply
    the 2-D DFT by sinc(ux w) sinc(uy w) exp(-2 pi^2 s^2 |u|^2) with u in cycles per pixel wrapped to (-1/2, 1/2], keep
    the real part of the inverse.  Pinned by tests/golden/smooth_pad.npz (the reference function itself, executed)."""
    npad = int(np.ceil(tophatwidth + 6 * gaussiansigma + 1))
    npad += (4 - npad) % 4
    ny, nx = image.shape
    big = np.zeros((ny + 2 * npad, nx + 2 * npad))
    big[npad : npad + ny, npad : npad + nx] = image

    def freq(n):
        u = np.arange(n) / n
        return np.where(u > 0.5, u - 1, u)

    uy, ux = freq(big.shape[0])[:, None], freq(big.shape[1])[None, :]
    h = np.sinc(ux * tophatwidth) * np.sinc(uy * tophatwidth) * np.exp(-2.0 * np.pi**2 * gaussiansigma**2 * (ux**2 + uy**2))
    return np.real(np.fft.ifft2(np.fft.fft2(big) * h))


# ---------------------------------------------------------------------------------------------------------------
# whole-stamp oracle: tables -> A, B -> LA kernel -> map post-processing -> coaddition, for a synthetic workload
# configuration `cfg` (pyimcom_amd.synth.WorkloadConfig: plain numbers, nothing of the product is executed)
def stamp_tables(cfg, psfs, target):
    """Geom, the padded table stack [E(E+1)/2 self tables + E input-output tables per target] and C per target
    (psfutil.py:943-986, 1178-1294)."""
    g = Geom(cfg.npixpsf, cfg.oversamp, cfg.dtheta_as / 3600.0, cfg.flat_penalty)
    r_in, r_out = pad_and_rfft2(psfs, g), pad_and_rfft2(target, g)
    tri = overlap_self(r_in, g)
    cross = overlap_cross(r_in, r_out, g)  # [E, n_out, ...] -> target-major stack
    io = np.concatenate([cross[:, o] for o in range(cross.shape[1])])
    Cs = overlap_out_C(r_out, g)
    tabs = np.concatenate([tri, io])
    return g, np.pad(tabs, ((0, 0), (6, 6), (6, 6))), np.asarray(Cs, dtype=np.float64)


def stamp_full(cfg, g, tables_pad, C, stamp, pair_tab, pair_pen, io_tab, timings=None):
    """One stamp through the reference's stamp driver: _build_system_matrices (coadd.py:1002-1122: A, -B/2, LA kernel,
    the Iterative clamp, the map taper) and _perform_coaddition (1294-1363).  ``timings``: optional dict that
    receives the seconds spent per stage (build / solve / epilogue; the Cholesky kernel adds factor and tri_solve, the two parts
    of solve)."""
    import time

    t0 = time.perf_counter()
    A, Bt = stamp_system(g, stamp.x, stamp.y, stamp.expo, tables_pad, pair_tab, pair_pen, io_tab, stamp.out_x0,
                         stamp.out_y0, cfg.n2f)
    mB = np.ascontiguousarray(Bt.T)
    t1 = time.perf_counter()
    if cfg.kernel in ("Iterative", "Empirical"):
        g1 = np.arange(cfg.n2f, dtype=np.float64)
        oy, ox = np.repeat(stamp.out_y0 + g1, cfg.n2f), np.tile(stamp.out_x0 + g1, cfg.n2f)
        if cfg.kernel == "Iterative":
            T, UC, Sigma, kappa, info = iter_kernel(A, mB, C, np.array(cfg.kappaC), cfg.uctarget, cfg.sigmamax, oy, ox,
                                                    stamp.y, stamp.x, cfg.rho, getattr(cfg, "iter_rtol", 1.5e-3), getattr(cfg, "iter_max", 30))
            UC, Sigma = iterative_clamp(UC, Sigma)
        else:
            T, UC, Sigma, kappa, info = empir_kernel(A, mB, C, np.array(cfg.kappaC), oy, ox, stamp.y, stamp.x, cfg.rho)
    else:
        if cfg.kernel == "Eigen":
            T, UC, Sigma, kappa, info = eigen_kernel(A, mB, C, np.array(cfg.kappaC), cfg.uctarget, cfg.sigmamax)
        else:
            T, UC, Sigma, kappa, info = chol_kernel(A, mB, C, np.array(cfg.kappaC), cfg.uctarget, cfg.sigmamax, timings=timings)
    s = (cfg.n2f, cfg.n2f)
    UC, Sigma, kappa = UC.reshape(s).copy(), Sigma.reshape(s).copy(), kappa.reshape(s).copy()
    if cfg.fade > 0:  # coadd.py:1118-1122
        for a in (kappa, Sigma, UC):
            trapezoid(a, cfg.fade)
    t2 = time.perf_counter()
    T3 = T[None].copy()
    outimage, Tsum_stamp, Tsum_inpix, Neff = perform_coaddition(T3, stamp.indata, stamp.expo, stamp.n_expo, cfg.n2f,
                                                                 cfg.n2, cfg.fade)
    t3 = time.perf_counter()
    if timings is not None:
        for k, v in (("build", t1 - t0), ("solve", t2 - t1), ("epilogue", t3 - t2)):
            timings[k] = timings.get(k, 0.0) + v
    return dict(A=A, Bt=Bt, T=T3[0], UC=UC, Sigma=Sigma, kappa=kappa, outimage=outimage[0], Tsum_stamp=Tsum_stamp[0],
                Tsum_inpix=Tsum_inpix[0], Neff=Neff[0], info=info)
