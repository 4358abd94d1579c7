"""Drop-in module named ``pyimcom_croutines``: put this directory on ``sys.path`` and the reference's
three-level import fallback (lakernel.py:41-47, psfutil.py:37-49, layer.py:44-50) picks the HIP routines
up with zero source changes when furry_parakeet is absent:

    PYTHONPATH=/path/to/repo:/path/to/repo/pyimcom_amd/shim python run_pyimcom.py cfg.json 0
"""

from pyimcom_amd.routines import (  # noqa: F401
    build_reduced_T_wrap,
    gridD5512C,
    iD5512C,
    iD5512C_getw,
    iD5512C_sym,
    lakernel1,
)
