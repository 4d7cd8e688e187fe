"""oracle/cv2_probe.py (the opportunistic real-OpenCV datum of SURVEY.md 8d(2)) exercised without OpenCV: a stand-in
`cv2` module backed by the oracle drives every line of the probe, a perturbed stand-in checks that the first differing
primitive is named, and a missing module must end in a clean "absent" report -- never an exception."""
import sys
import types

import numpy as np
import pytest


def _fake_cv2(perturb=None):
    import orc
    m = types.ModuleType("cv2")
    m.__version__ = "0.0-oracle-standin"
    m.INTER_LINEAR, m.COLOR_BGR2GRAY, m.CV_32S, m.CV_64F = 1, 6, 4, 6

    class CascadeClassifier:
        def __init__(self, path):
            self.c = orc.load_cascade(path)

        def empty(self):
            return False

        def detectMultiScale(self, img, scaleFactor=1.1, minNeighbors=3, flags=0, minSize=(0, 0)):
            r = orc.detect_multiscale(self.c, img, scaleFactor, minNeighbors, flags, minSize)
            if perturb == "detect" and len(r):
                r = r.copy(); r[0, 0] += 1
            return r[::-1]                      # a different order: the probe compares sets

    def equalize(img):
        out = orc.equalize_hist(img)
        if perturb == "equalizeHist":
            out = out.copy(); out[0, 0] ^= 1
        return out

    m.CascadeClassifier = CascadeClassifier
    m.resize = lambda img, size, interpolation=1: orc.resize_linear(img, size[0], size[1])
    m.cvtColor = lambda img, code: orc.bgr2gray(img)
    m.equalizeHist = equalize
    m.integral2 = lambda img, sdepth=4, sqdepth=6: orc.integral(img)
    m.getNumThreads = lambda: 1
    return m


@pytest.fixture()
def frames():
    from nubovca import synth
    return [synth.make_bgr(320, 240, 600 + i, "natural", [(60 + 10 * i, 40, 120)]) for i in range(3)]


def test_probe_with_faithful_standin(monkeypatch, synth_xml, frames):
    import cv2_probe
    monkeypatch.setitem(sys.modules, "cv2", _fake_cv2())
    r = cv2_probe.probe(synth_xml, frames, width_to_process=160, scale_factor=1.25)
    assert r["available"] and r["frames"] == 3 and r["fps"] > 0
    assert r["boxes_equal"] is True and r["first_difference"] is None
    assert all(r["primitives_equal"].values()) and set(r["primitives_equal"]) == {"resize", "cvtColor", "equalizeHist", "integral"}
    r = cv2_probe.probe(synth_xml, frames, width_to_process=0, scale_factor=1.2)          # full-resolution mode: no resize
    assert r["boxes_equal"] is True and r["primitives_equal"]["resize"]


@pytest.mark.parametrize("what", ["equalizeHist", "detect"])
def test_probe_names_the_first_difference(monkeypatch, synth_xml, frames, what):
    import cv2_probe
    monkeypatch.setitem(sys.modules, "cv2", _fake_cv2(perturb=what))
    r = cv2_probe.probe(synth_xml, frames, width_to_process=0, scale_factor=1.2)
    assert r["available"]
    if what == "equalizeHist":
        assert r["primitives_equal"]["equalizeHist"] is False and r["first_difference"] == "equalizeHist"
        assert r["primitives_equal"]["cvtColor"] and r["primitives_equal"]["resize"]
    else:
        assert all(r["primitives_equal"].values())
        assert r["boxes_equal"] is False and r["first_difference"].startswith("detectMultiScale")


def test_probe_without_cv2_is_a_clean_report(monkeypatch, synth_xml, frames):
    import cv2_probe
    monkeypatch.setitem(sys.modules, "cv2", None)           # import cv2 -> ImportError
    r = cv2_probe.probe(synth_xml, frames)
    assert r["available"] is False and "absent" in r["note"]


def test_probe_survives_a_broken_install(monkeypatch, synth_xml, frames):
    import cv2_probe
    broken = types.ModuleType("cv2")                         # imports, but nothing works
    monkeypatch.setitem(sys.modules, "cv2", broken)
    r = cv2_probe.probe(synth_xml, frames)
    assert r["available"] and r["boxes_equal"] is None and "probe failed" in r["first_difference"]
