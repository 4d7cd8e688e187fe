"""A short burst of the randomised parity tool (tests/fuzz_parity.py) with a fixed seed."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_fuzz_burst():
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "fuzz_parity.py"), "8", "12345"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "no mismatch" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
