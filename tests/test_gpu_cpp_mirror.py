"""GPU: the reference's own in-file tests (src/dsp.rs:57-83, 136-157) written in C++ against the C++ mirror of
`doppler::dsp` (include/doppler_dsp.hpp, over the C ABI), checked bit for bit against the oracle."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_mirror_runs_reference_tests():
    exe = os.path.join(ROOT, "tests", "cpp", "test_dsp")
    assert os.path.exists(exe), "build with `make cpptest`"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "all passed" in r.stdout
