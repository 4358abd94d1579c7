"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol that
include/imcom_hip.h declares; no compute calls here (no GPU in the build container)."""

import os
import re

import pytest

from tests.conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "imcom_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(imcom_[A-Za-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    import __graft_entry__ as g

    g.build()
    from pyimcom_amd import _lib

    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(_lib.lib, n), f"{n} declared in include/imcom_hip.h but not exported"
    assert set(names) == set(_lib.EXPORTED), set(names) ^ set(_lib.EXPORTED)
    assert _lib.lib.imcom_version() == 100


def test_no_gpu_fails_loudly():
    """Without a usable gfx950 device the context cannot be created and there is no fallback."""
    import pytest

    from pyimcom_amd import _lib

    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.ImcomError):
        _lib.Context(0)


def test_product_does_not_import_oracle():
    """Nothing under pyimcom_amd/ may import or call the oracle or the test-side checker (tests/parity.py)."""
    pkg = os.path.join(ROOT, "pyimcom_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py") and not dirpath.endswith(os.path.join("csrc", "tools")):  # tools/: developer scripts, not shipped code
                src = open(os.path.join(dirpath, f)).read()
                assert "from oracle" not in src and "import oracle" not in src and "tests" not in [w for l in src.splitlines()
                        if l.startswith(("from ", "import ")) for w in l.replace(".", " ").split()[1:2]], f


def test_package_import_asks_for_eight_hardware_queues():
    """The HIP runtime reads GPU_MAX_HW_QUEUES at its first call; the package sets 8 on import unless the caller has chosen (on the default
    of four, streams that share a queue wait for each other: profiles/r04_negative_results.txt item 18)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import os; import pyimcom_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    env["PYTHONPATH"] = root
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120).stdout.strip() == "8"
    env["GPU_MAX_HW_QUEUES"] = "6"
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120).stdout.strip() == "6"


@pytest.mark.gpu
def test_import_after_the_runtime_is_up_warns_and_changes_nothing():
    """A plug-in does not reconfigure its host's HIP runtime behind its back: when the device has been touched before the import
    (GPU_MAX_HW_QUEUES can no longer take effect) the package leaves the environment alone and says so once."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, warnings, torch; torch.zeros(1, device='cuda:0'); warnings.simplefilter('always');\n"
            "with warnings.catch_warnings(record=True) as w:\n    import pyimcom_amd\n"
            "print('GPU_MAX_HW_QUEUES' in os.environ, sum('GPU_MAX_HW_QUEUES' in str(x.message) for x in w))")
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    env["PYTHONPATH"] = root
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.stdout.split() == ["False", "1"], (out.stdout, out.stderr[-500:])
