import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "nubomedia-vca_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _abort_evidence():
    """An abort() inside a GPU test run used to leave nothing but "Fatal Python error: Aborted": glibc writes its fatal messages
    (heap checks, stack protector) to the controlling terminal unless told otherwise, and faulthandler shows Python frames only.
    Make both visible: messages to stderr, and the C call stack of the raising thread (tests/san/abrt_trace.c)."""
    os.environ.setdefault("LIBC_FATAL_STDERR_", "1")
    try:
        import ctypes
        import subprocess
        src = os.path.join(ROOT, "tests", "san", "abrt_trace.c")
        out = os.path.join(ROOT, "tests", "san", "build", "abrt_trace.so")
        if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
            os.makedirs(os.path.dirname(out), exist_ok=True)
            subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-g", "-o", out, src, "-ldl"], check=True, capture_output=True, timeout=120)
        lib = ctypes.CDLL(out)
        lib.abrt_trace_fd = os.dup(2)          # taken while the runner configures, as its own fault handler does
        lib.abrt_trace_install(lib.abrt_trace_fd)
        return lib
    except Exception:          # no compiler, read-only tree: the tests run without the extra evidence
        return None


_ABRT = None


def pytest_configure(config):
    global _ABRT
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _ABRT = _abort_evidence()


def pytest_sessionstart(session):
    # the fault handler plugin installs its own SIGABRT handler while configuring: go on top of it (it is chained to)
    if _ABRT is not None:
        _ABRT.abrt_trace_install(_ABRT.abrt_trace_fd)


@pytest.fixture(scope="session")
def synth_xml():
    from nubovca import synth
    return synth.synthetic_cascade_xml()


@pytest.fixture(scope="session")
def calibrated_xml():
    """bench.py's headline cascade: stage thresholds calibrated on the bench's own content (every early stage rejects about half)"""
    from nubovca import synth
    return synth.calibrated_cascade_xml()


@pytest.fixture(scope="session")
def small_xml():
    """6-stage cascade: fast enough for exhaustive CPU checks."""
    from nubovca import synth
    return synth.synthetic_cascade_xml(seed=7, stages=[3, 8, 12, 16, 20, 24])


@pytest.fixture(scope="session")
def orc_cascade(synth_xml):
    import orc
    return orc.parse_cascade_xml(synth_xml)


@pytest.fixture(scope="session")
def orc_small(small_xml):
    import orc
    return orc.parse_cascade_xml(small_xml)
