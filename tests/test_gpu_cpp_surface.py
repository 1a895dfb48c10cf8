"""C++-level checks of include/gunrock/ that the C ABI does not reach (tests/cpp/engine_tests.hip):
frontier_t methods, parallel_for, enactor-overload swap rules, explicit-frontier advance for every
schedule, batch, rejection of unsupported variants -- and the same binary built with
-DGRX_ADVANCE_LB_OVERRIDE=bucketing; boundary_tests.hip covers the
reference include paths beside the hot path (launch_box, array, sample, smtx, print, memory)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


@pytest.mark.parametrize("binary", ["engine_tests", "engine_tests_override", "boundary_tests"])
def test_cpp_engine_tests(binary):
    path = os.path.join(ROOT, "tests", "cpp", binary)
    if not os.path.exists(path):
        from essentials_amd.build import build_cpp_tests
        build_cpp_tests()
    r = subprocess.run([path], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "all passed" in r.stdout
