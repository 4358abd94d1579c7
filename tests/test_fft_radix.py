"""The in-register butterflies of the line FFTs (pyimcom_amd/csrc/fft_radix.h) compile for the host: every radix, forward
and inverse, against the direct DFT sum (tests/native/fft_radix_check.cpp, g++)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_butterflies_vs_direct_sum(tmp_path):
    exe = tmp_path / "fft_radix_check"
    subprocess.check_call(["g++", "-O1", "-I", os.path.join(ROOT, "pyimcom_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "fft_radix_check.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    rows = [ln.split() for ln in out.stdout.strip().splitlines()]
    assert sorted({int(r[0]) for r in rows}) == [2, 3, 4, 5, 8, 16] and len(rows) == 12
    assert all(float(r[2]) < 2e-15 for r in rows), rows
