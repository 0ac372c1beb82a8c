"""CPU sanitizer build (AddressSanitizer + UBSan) of the host-side MT19937 stream code -- the GF(2) jump (mt19937.cpp) and
the regeneration plan of a device sub-batch (mtplan.cpp) -- run through its own driver (tests/native/sanitize_mt.cpp).
Sanitizers run on the CPU build only; no GPU involved."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_mt19937_host_code_under_asan_ubsan():
    csrc = os.path.join(ROOT, "adaptive_matrix_solver_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "asan"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "sanitize_mt: all checks passed" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
