import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "nubomedia-vca_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def synth_xml():
    from nubovca import synth
    return synth.synthetic_cascade_xml()


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
