"""Batched symmetric eigendecomposition on the GPU (replaces numpy.linalg.eigh at lakernel.py:162,201,266)."""

import ctypes as C

import numpy as np

from ._lib import MEM_HOST, check, default_context, lib


def eigh(A, ctx=None):
    """Eigenvalues (ascending) and eigenvectors (columns) of a symmetric float64 matrix or a stack of them."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    single = A.ndim == 2
    if single:
        A = A[None]
    b, n, n2 = A.shape
    if n != n2:
        raise ValueError("eigh: square matrices expected")
    lam = np.zeros((b, n))
    Q = np.zeros((b, n, n))
    ns = np.full((b,), n, dtype=np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    if n > 0:
        check(lib.imcom_eigh((ctx or default_context()).handle, b, p(ns), n, p(A), p(lam), p(Q), MEM_HOST))
    return (lam[0], Q[0]) if single else (lam, Q)
