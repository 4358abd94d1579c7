"""ctypes binding of libimcom_hip.so (the C-ABI declared in include/imcom_hip.h).

The HIP library is the product: there is no CPU fallback.  If the shared object is missing or no
gfx950 device is usable, importing this module or creating a context raises -- loudly.
"""

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# IMCOM_HIP_LIB: another build of the same library (the developer build with its cross-check kernels, make DEV=1)
LIB_PATH = os.environ.get("IMCOM_HIP_LIB") or os.path.join(_HERE, "lib", "libimcom_hip.so")

MEM_HOST = 0
MEM_DEVICE = 1

PAIR_SWAP = 1 << 29  # build_A pair code bits (csrc/interp.hip)
PAIR_FLIP = 1 << 30


class ImcomError(RuntimeError):
    """A libimcom_hip call returned a negative status."""

    def __init__(self, status, msg):
        super().__init__(f"libimcom_hip error {status}: {msg}")
        self.status = status


class TableGeom(C.Structure):
    _fields_ = [("nsamp", C.c_int), ("nc", C.c_double), ("dscale", C.c_double), ("flat_penalty", C.c_double)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' at the repo root, or make -C pyimcom_amd/csrc). "
            "pyimcom_amd has no CPU fallback."
        )
    # torch ships its own HIP runtime under the same SONAME (libamdhip64.so.7).  One process must hold
    # exactly one HIP runtime, so torch's has to be the first one loaded; libimcom_hip then binds to it.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    return C.CDLL(LIB_PATH)


lib = _load()

_vp, _i, _l, _d = C.c_void_p, C.c_int, C.c_long, C.c_double
_ip = C.POINTER(C.c_int)

# name -> argtypes; every function returns int status except the two noted below
SIGNATURES = {
    "imcom_device_count": [_ip],
    "imcom_ctx_create": [_i, C.POINTER(_vp)],
    "imcom_ctx_destroy": [_vp],
    "imcom_ctx_set_stream": [_vp, _vp],
    "imcom_ctx_sync": [_vp],
    "imcom_ctx_workspace_bytes": [_vp, C.POINTER(C.c_size_t)],
    "imcom_ctx_workspace_release": [_vp],
    "imcom_ctx_set_workspace": [_vp, _vp, C.c_size_t],
    "imcom_ctx_workspace_needed": [_vp, C.POINTER(C.c_size_t)],
    "imcom_ctx_set_repair_hint": [_vp, _d],
    "imcom_ctx_set_repair_expect": [_vp, _i],
    "imcom_ctx_last_repair": [_vp, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "imcom_ctx_profile_enable": [_vp, _i],
    "imcom_ctx_profile_reset": [_vp],
    "imcom_ctx_profile_get": [_vp, C.c_char_p, C.POINTER(_d), C.POINTER(_l)],
    "imcom_ctx_mfma_probe": [_vp, _d, C.POINTER(_d)],
    "imcom_ctx_gemm_probe": [_vp, _i, _i, _i, _i, _i, _i, C.POINTER(_d)],
    "imcom_d5512_getw": [_vp, _vp, _l, _vp, _i],
    "imcom_interp_d5512": [_vp, _vp, _i, _i, _i, _vp, _vp, _l, _vp, _i, _i],
    "imcom_grid_d5512": [_vp, _vp, _i, _i, _vp, _vp, _l, _i, _i, _vp, _i],
    "imcom_lakernel1": [_vp, _vp, _vp, _l, _l, _d, _d, _d, _d, _i, _vp, _vp, _vp, _vp, _d, _i],
    "imcom_build_reduced_T": [_vp, _vp, _vp, _vp, _vp, _i, _l, _d, _d, _vp, _vp, _vp, _vp, _i],
    "imcom_solve_chol": [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _d, _d, _vp, _vp, _vp, _vp, _vp, _i],
    "imcom_solve_chol_stamps": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _d, _d, _vp, _vp, _vp, _vp, _vp],
    "imcom_solve_eigen": [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _d, _d, _i, _vp, _vp, _vp, _vp, _vp, _i],
    "imcom_solve_eigen_resident": [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _d, _d, _i, _vp, _vp, _vp, _vp, _vp],
    "imcom_solve_iter": [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _d, _d, _vp, _vp, _vp, _d, _d, _i, _i, _vp, _vp, _vp, _vp, _i],
    "imcom_solve_iter_stats": [_vp, _vp, _vp, _l],
    "imcom_solve_empir": [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _d, _vp, _vp, _vp, _d, _i, _vp, _vp, _vp, _vp, _i],
    "imcom_eigh": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i],
    "imcom_band_reduce": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i],
    "imcom_partition_pixels": [_vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _d, _d, _i, _vp, _vp, _vp, _vp, _vp],
    "imcom_select_pixels": [_vp, _i, _vp, _vp, _vp, _l, _i, _vp, _vp, _i, _vp, _vp, _vp, _d, _i, _vp, _vp, _vp, _vp, _vp, _i],
    "imcom_build_A": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, C.POINTER(TableGeom), _vp, _vp, _i, _vp],
    "imcom_build_B": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, C.POINTER(TableGeom), _vp, _i, _vp, _vp, _i, _i, _vp],
    "imcom_solve_chol_resident": [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _d, _d, _vp, _vp, _vp, _vp, _vp],
    "imcom_solve_chol_resident_begin": [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _d, _d, _vp, _vp, _vp, _vp],
    "imcom_solve_chol_resident_end": [_vp, _i, _vp],
    "imcom_solve_chol_resident_redo": [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _d, _d, _vp, _vp, _vp, _vp, _vp, _vp],
    "imcom_coadd_epilogue": [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp],
    "imcom_solve_chol_resident_coadd": [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _d, _d, _vp, _vp, _vp, _vp, _vp,
                                        _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp],
    "imcom_trapezoid_f32": [_vp, _vp, _l, _i, _i],
    "imcom_clamp_min_f32": [_vp, _vp, _l, C.c_float],
    "imcom_sample_psf": [_vp, _i, _vp, _i, _i, _vp, _i, _i, _i, _vp, _i],
    "imcom_lattice_positions": [_vp, _i, _i, _vp, _vp, _i, _vp, _i],
    "imcom_psf_gaussian": [_vp, _i, _d, _d, _vp, _i],
    "imcom_psf_simple_airy": [_vp, _i, _d, _d, _d, _d, _vp, _i],
    "imcom_smooth_and_pad": [_vp, _i, _vp, _i, _i, _d, _d, _vp, _i],
    "imcom_psf_overlap": [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _i, _vp, _vp],
    "imcom_psf_spectra": [_vp, _vp, _i, _i, _i, _vp],
    "imcom_psf_overlap_spectra": [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _i, _vp, _vp],
    "imcom_psf_overlap_spectra_win": [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp],
    "imcom_psf_overlap_spectra_slots": [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp],
    "imcom_block_accumulate": [_vp, _i, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _i],
    "imcom_solve_eigen_workspace": [_i, _i, _i, _i, _vp],
    "imcom_solve_chol_workspace": [_i, _i, _i, _i, _i, _vp],
    "imcom_block_place": [_vp, _i, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _i],
    "imcom_block_combine": [_vp, _i, _i, _i, _l, _vp, _i, _vp, _i, _i, _i, _i],
    "imcom_compress_map_f32": [_vp, _vp, _l, _i, _i, _vp],
    "imcom_trapezoid_recover_f32": [_vp, _vp, _l, _i, _i, _i, _i, _i, _i, _i],
}
_cdll = lib
for _name, _args in SIGNATURES.items():
    _f = getattr(_cdll, _name)
    _f.argtypes = _args
    _f.restype = C.c_int

IMCOM_ERR_NOMEM = -3
_CONTEXTS = {}  # handle value -> Context (weak): the contexts whose workspace this module provides from torch's allocator


def _with_workspace(name, fn):
    """A library call on a context whose device workspace this module owns (Context, below): when the call reports that the
    workspace it was given is too small (IMCOM_ERR_NOMEM before anything was queued), the context's buffer is replaced by one of
    the size the call asked for -- taken from torch's allocator, the only allocator on the device -- and the call is made again."""

    def call(*args):
        rc = fn(*args)
        if rc != IMCOM_ERR_NOMEM or not args:
            return rc
        h = args[0]
        ref = _CONTEXTS.get(h.value if isinstance(h, C.c_void_p) else h)
        ctx = ref() if ref is not None else None
        if ctx is None or not ctx._grow_to_needed():
            return rc
        return fn(*args)

    call.__name__ = name
    return call


class _Lib:
    """The shared library with every context-taking entry wrapped by ``_with_workspace``; everything else passes through."""

    def __init__(self, cdll):
        self._cdll = cdll
        for name, args in SIGNATURES.items():
            f = getattr(cdll, name)
            takes_ctx = bool(args) and args[0] is _vp and name not in ("imcom_ctx_destroy", "imcom_ctx_set_workspace", "imcom_ctx_workspace_needed",
                                                                       "imcom_ctx_workspace_bytes", "imcom_ctx_workspace_release", "imcom_ctx_set_repair_hint", "imcom_ctx_last_repair")
            setattr(self, name, _with_workspace(name, f) if takes_ctx else f)

    def __getattr__(self, name):  # (only what __init__ did not set)
        return getattr(self._cdll, name)


lib = _Lib(_cdll)
_cdll.imcom_version.argtypes = []
_cdll.imcom_version.restype = C.c_int
_cdll.imcom_dev_build.argtypes = []
_cdll.imcom_dev_build.restype = C.c_int
_cdll.imcom_psf_spectra_size.argtypes = [_i, _i]
_cdll.imcom_psf_spectra_size.restype = C.c_long
_cdll.imcom_smooth_pad_width.argtypes = [_d, _d]
_cdll.imcom_smooth_pad_width.restype = C.c_int
_cdll.imcom_last_error.argtypes = []
_cdll.imcom_last_error.restype = C.c_char_p

EXPORTED = sorted(list(SIGNATURES) + ["imcom_version", "imcom_dev_build", "imcom_last_error", "imcom_psf_spectra_size", "imcom_smooth_pad_width"])


def source_sha16():
    """First 16 hex digits of the SHA-256 over the kernel sources (csrc/*.hip, *.h, the C-ABI header): stamps measurements
    (profiles/r*_pmc_traffic.json) with the build they were taken on."""
    import glob
    import hashlib

    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h"))
                   + glob.glob(os.path.join(_HERE, "..", "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def check(status):
    if status != 0:
        raise ImcomError(status, lib.imcom_last_error().decode("utf-8", "replace"))


def device_count():
    n = C.c_int(0)
    st = lib.imcom_device_count(C.byref(n))
    return n.value if st == 0 else 0


class Context:
    """One libimcom context = one GPU + one stream + one workspace (not thread-safe).

    Device memory has ONE owner: torch's caching allocator.  The context's workspace is a torch tensor handed to the library
    (imcom_ctx_set_workspace): the library never allocates device memory itself; a call that needs a larger workspace says so,
    the tensor is replaced by one of exactly that size (``_grow_to_needed``; the old one goes back to torch first) and the call is
    made again.  ``torch.cuda.memory_allocated`` therefore accounts for everything on the device, and a planner that reads torch's
    numbers (pyimcom_amd.blockrun) plans with exact figures.  ``own_workspace=False`` (or no torch): the library's own hipMalloc."""

    def __init__(self, device=0, own_workspace=True):
        self._h = _vp()
        check(lib.imcom_ctx_create(int(device), C.byref(self._h)))
        self.device = int(device)
        self._ws = None
        self._owns_ws = False
        if own_workspace:
            try:
                import torch  # noqa: F401

                check(lib.imcom_ctx_set_workspace(self._h, None, 0))
                self._owns_ws = True
                import weakref

                _CONTEXTS[self._h.value] = weakref.ref(self)
            except ImportError:  # pragma: no cover
                pass

    def _grow_to_needed(self):
        """Replace the workspace tensor by one of the size the last call asked for.  False: nothing to do (the call's failure was
        not about this workspace)."""
        if not self._owns_ws or not self._h:
            return False
        import torch

        need = self.workspace_needed()
        have = 0 if self._ws is None else self._ws.numel()
        if need <= have:
            return False
        dev = torch.device("cuda", self.device)
        torch.cuda.synchronize(dev)  # whatever still runs in the old buffer
        check(lib.imcom_ctx_set_workspace(self._h, None, 0))
        self._ws = None
        torch.cuda.empty_cache()  # the old buffer goes back to the driver: a larger one cannot be carved out of it
        size = (need + (1 << 21) - 1) >> 21 << 21
        try:
            self._ws = torch.empty(size, dtype=torch.uint8, device=dev)
        except torch.OutOfMemoryError as e:
            raise ImcomError(IMCOM_ERR_NOMEM, f"device workspace of {size} bytes: {e}") from None
        check(lib.imcom_ctx_set_workspace(self._h, _vp(self._ws.data_ptr()), size))
        return True

    def set_repair_hint(self, lmin_abs):
        """An estimate of max |w[0]| for the stamps the next Cholesky calls repair (lakernel.py:262-279); 0 / None clears it
        (imcom_ctx_set_repair_hint: the smallest-eigenvalue iteration then starts close to its answer)."""
        check(lib.imcom_ctx_set_repair_hint(self.handle, float(lmin_abs or 0.0)))

    def set_repair_expect(self, expect):
        """The host-array Cholesky entries go straight to the repair on the next calls (imcom_ctx_set_repair_expect)."""
        check(lib.imcom_ctx_set_repair_expect(self.handle, int(bool(expect))))

    def last_repair(self):
        """(stamps the last Cholesky call repaired, smallest and largest w[0] among them)."""
        c, a, b = C.c_int(0), C.c_double(0.0), C.c_double(0.0)
        check(lib.imcom_ctx_last_repair(self.handle, C.byref(c), C.byref(a), C.byref(b)))
        return c.value, a.value, b.value

    def workspace_needed(self):
        b = C.c_size_t(0)
        check(lib.imcom_ctx_workspace_needed(self.handle, C.byref(b)))
        return b.value

    @property
    def handle(self):
        if not self._h:
            raise ImcomError(-1, "context already destroyed")
        return self._h

    def set_stream(self, stream_ptr):
        check(lib.imcom_ctx_set_stream(self.handle, _vp(stream_ptr) if stream_ptr else None))

    def sync(self):
        check(lib.imcom_ctx_sync(self.handle))

    def workspace_bytes(self):
        b = C.c_size_t(0)
        check(lib.imcom_ctx_workspace_bytes(self.handle, C.byref(b)))
        return b.value

    def release_workspace(self):
        """Hand the device workspace back (it otherwise keeps the size of the largest call made on this context)."""
        check(lib.imcom_ctx_workspace_release(self.handle))
        if self._owns_ws:
            self._ws = None  # (imcom_ctx_workspace_release has drained the context's streams)

    def profile_enable(self, on=True):
        """on = 2: also the per-launch scopes inside long stages (the band reduction's "symv4")."""
        check(lib.imcom_ctx_profile_enable(self.handle, int(on)))

    def profile_reset(self):
        check(lib.imcom_ctx_profile_reset(self.handle))

    def profile_get(self, family):
        ms, n = _d(0.0), _l(0)
        check(lib.imcom_ctx_profile_get(self.handle, family.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def iter_stats(self, nsteps=None):
        """What the last imcom_solve_iter call on this context did at its last kappa node (imcom_solve_iter_stats): a dict -- patches of
        4 x 4 output pixels, flops / bytes of their conjugate-gradient steps, the largest union selection, whether the blocked solver ran
        -- and, with ``nsteps`` = batch * m of that call, the CG steps used per output pixel (int32 [nsteps])."""
        import numpy as np

        st = (_d * 8)()
        steps = None if nsteps is None else np.zeros(int(nsteps), dtype=np.int32)
        check(lib.imcom_solve_iter_stats(self.handle, st, None if steps is None else _vp(steps.ctypes.data), 0 if steps is None else int(nsteps)))
        out = {"patches": int(st[0]), "up2_steps": float(st[1]), "patch_steps": float(st[2]), "up2": float(st[3]), "max_union": int(st[4]),
               "blocked": bool(st[5]), "flops": 32.0 * float(st[1]), "bytes": float(st[6]) if st[7] else 8.0 * float(st[1]), "bytes_full_storage": 8.0 * float(st[1]),
               "half_storage": bool(st[7])}  # (bytes: what the patches streamed over their steps -- the whole sub-matrix per step, or its lower tiles)
        return (out, steps) if nsteps is not None else out

    def mfma_probe(self, millis=50.0):
        """Rate [TFLOP/s] of a pure fp64 MFMA loop on every SIMD (imcom_ctx_mfma_probe): the chip's ceiling under matrix load."""
        tf = _d(0.0)
        check(lib.imcom_ctx_mfma_probe(self.handle, float(millis), C.byref(tf)))
        return tf.value

    def gemm_probe(self, variant, M=2304, N=2304, K=2304, batch=8, reps=5):
        """TFLOP/s of the tile engine's k loop as a plain batched product (imcom_ctx_gemm_probe): variant 0 = 128 x 128 tiles, 1 = 256 x 128."""
        tf = _d(0.0)
        check(lib.imcom_ctx_gemm_probe(self.handle, int(variant), int(M), int(N), int(K), int(batch), int(reps), C.byref(tf)))
        return tf.value

    def close(self):
        if self._h:
            _CONTEXTS.pop(self._h.value, None)
            lib.imcom_ctx_destroy(self._h)
            self._h = _vp()
            self._ws = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default = {}
_lock = threading.Lock()


def default_context(device=0):
    """Process-wide context per device (created on first use; raises without a usable gfx950)."""
    with _lock:
        ctx = _default.get(device)
        if ctx is None:
            ctx = _default[device] = Context(device)
        return ctx
