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


def band_reduce(A, n=None, ctx=None):
    """Householder reduction to band form (bandwidth 4), the basis of the Eigen kernel's kappa search: A [b, ld, ld] (ld a
    multiple of 128; the leading n[s] x n[s] of every matrix is used) -> band [b, 5, ld] with band[t][i] = B[i+t][i], the
    reflectors V [b, ld, ld] (row r = v_r, pivot at r + 4) and tau [b, ld]; A = Q B Q^T, Q = H_0 H_1 ..."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    b, ld, _ = A.shape
    ns = np.full((b,), ld, dtype=np.int32) if n is None else np.ascontiguousarray(n, dtype=np.int32)
    band, V, tau = np.zeros((b, 5, ld)), np.zeros((b, ld, ld)), np.zeros((b, ld))
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    check(lib.imcom_band_reduce((ctx or default_context()).handle, b, p(ns), ld, p(A), p(band), p(V), p(tau), MEM_HOST))
    return band, V, tau
