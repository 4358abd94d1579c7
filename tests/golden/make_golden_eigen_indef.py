#!/usr/bin/env python3
"""Golden vectors for the Eigen kernel on system matrices that are NOT positive semi-definite, by RUNNING the reference.

The reference's EigenKernel (lakernel.py:141-223) diagonalises A with numpy.linalg.eigh and divides by lam + kappa whatever
its sign (single kappa 154-172; multi kappa through routine.lakernel1, routine.py:341-430): it is the kernel a user turns
to when the Cholesky factorisation fails, so it must serve matrices with eigenvalues below -kappa.  Recorded here: the
reference's outputs on

  repair   the 6 x 6 cosine system of tests/pyimcom/test_la.py minus 1e-3 I (lakernel.npz: repair_A; lam_min = -1e-3)
  gau      the Gaussian system N = 169, m = 81, two target PSFs (lakernel.npz: gau_A) minus `shift` I
  chain    the real PSF-overlap system of one output stamp, N = 220, m = 196, built by the reference's own PSFOvl / SysMatA
           chain (stamp_chain_mid.npz: A, mBhalf, C) minus `shift` I: half of its eigenvalues negative

each with one kappa node (A + kappa I indefinite) and with several (eigenvalues inside and below the bracket).  The inputs
are those of the existing fixtures; this file holds the shifts, the kappa nodes and the reference's outputs only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_eigen_indef.py
"""

import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _load_reference, run_kernel  # noqa: E402


def pole_distance(A, mB, C, kC, nbis=13, uct=1e-6, smax=0.5):
    """min |lam_i + kappa| over every kappa the reference evaluates (numpy restatement of the bisection, diagnostics only)."""
    lam, Q = np.linalg.eigh(A)
    best = np.inf
    for k in range(mB.shape[0]):
        P = mB[k] @ Q
        kmin, kmax = kC[0] * C[k], kC[-1] * C[k]
        if len(kC) == 1:
            best = min(best, np.abs(lam + kmin).min())
            continue
        for a in range(P.shape[0]):
            factor, kap = np.sqrt(kmax / kmin), np.sqrt(kmax * kmin)
            for _ in range(nbis + 1):
                best = min(best, np.abs(lam + kap).min())
                var = P[a] / (lam + kap)
                s2, s1 = np.sum(var * var), np.sum((lam + 2 * kap) * var * var)
                factor = np.sqrt(factor)
                kap *= 1.0 / factor if (1 - s1 / C[k] > uct and s2 < smax) else factor
    return best


def main():
    _, lakernel, _ = _load_reference()
    la = np.load(f"{HERE}/lakernel.npz")
    ch = np.load(f"{HERE}/stamp_chain_mid.npz")
    out = {}
    cases = (
        # name, A, mBhalf, C, n2f, shift, single kappa/C, multi kappa/C, uctarget, sigmamax
        ("repair", la["repair_A"], la["cos_mBhalf"], np.atleast_1d(la["cos_C"]), 4, 0.0, [1e-4 / float(la["cos_C"])], [1e-4, 1e-3, 1e-2], 1e-4, 1.0),
        ("gau", la["gau_A"], la["gau_mBhalf"], la["gau_C"], 9, 2e-3, [6e-4], [1e-5, 1e-4, 1e-3], 1e-6, 0.5),
        ("chain", ch["A"], ch["mBhalf"], ch["C"], 14, 1e-4, [2e-3], [1e-4, 1e-2, 1e-1], 1e-6, 0.5),
    )
    for name, A0, mB, C, n2f, shift, k1, km, uct, smax in cases:
        A = A0 - shift * np.identity(A0.shape[0])
        lam = np.linalg.eigvalsh(A)
        out[f"{name}_shift"], out[f"{name}_uctarget"], out[f"{name}_sigmamax"] = shift, uct, smax
        for tag, kC in (("eig1", np.array(k1)), ("eigm", np.array(km))):
            kap_lo = kC[0] * float(np.min(C))
            assert lam[0] + kap_lo < 0, (name, tag, lam[0], kap_lo)  # A + kappa I is indefinite at the lowest node
            r = run_kernel(lakernel.EigenKernel, A, mB, C, n2f, kC, uct, smax)
            assert all(np.isfinite(v).all() for v in r.values())
            out[f"{name}_{tag}_kappaC"] = kC
            for k, v in r.items():
                out[f"{name}_{tag}_{k}"] = v
            print(f"{name} {tag}: N = {A.shape[0]}, lam_min = {lam[0]:.3e}, negative eigenvalues {int((lam < 0).sum())}, below -kappa_lo "
                  f"{int((lam < -kap_lo).sum())}, min |lam + kappa| over the evaluated kappas = {pole_distance(A, mB, C, kC, uct=uct, smax=smax):.2e}, "
                  f"|T| max {np.abs(r['T']).max():.3e}, UC {r['UC'].min():.3e}..{r['UC'].max():.3e}, Sigma max {r['Sigma'].max():.3e}")
    np.savez_compressed(f"{HERE}/eigen_indef.npz", **out)


if __name__ == "__main__":
    main()
