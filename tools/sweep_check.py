"""Ad-hoc robustness sweep of the resident path against the oracle over unusual small configurations."""
import dataclasses
import sys

sys.path.insert(0, ".")
from pyimcom_amd import synth  # noqa: E402
from tests import parity as smoke  # noqa: E402

base = synth.CONFIGS["tiny"]
variants = {
    "one_frame": dict(n_inframe=1),
    "one_expo": dict(n_expo=1),
    "odd_n2_nofade": dict(n2=7, fade=0),
    "fade3": dict(fade=3),
    "wide_pad": dict(inpad_as=0.3),
    "multi_kappa4": dict(kappaC=(1e-5, 1e-4, 1e-3, 1e-2)),
    "multi_kappa5": dict(kappaC=(1e-6, 1e-5, 1e-4, 1e-3, 1e-2), _tolT=10.0),
    "airy": dict(psf="airy"),
    "no_penalty": dict(flat_penalty=0.0),
    "eigen_multi": dict(kernel="Eigen", kappaC=(1e-5, 1e-2)),
    "empirical": dict(kernel="Empirical"),
    "two_targets_eigen": dict(kernel="Eigen", n_out=2),
    "many_expo": dict(n_expo=9),
}
bad = 0
for name, kw in variants.items():
    kw = dict(kw)
    scale = kw.pop("_tolT", 1.0)
    cfg = dataclasses.replace(base, name=name, **kw)
    try:
        rep = smoke.check_batch(cfg, n_stamps=3, tolT_scale=scale)
        k = [v for kk, v in rep.items() if kk.startswith("stamp")][0]
        print(f"{name:20s} ok   n={k['n']:4d} A={k['A']:.1e} B={k['B']:.1e} T={k['T']:.1e}")
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(f"{name:20s} FAIL {type(e).__name__}: {str(e)[:300]}")
print("failures:", bad)
sys.exit(1 if bad else 0)
