"""HIP path against REAL OpenCV -- only if a cv2 module already exists on the box (nothing is ever installed; the build
container and, as far as known, the GPU boxes have none, in which case this file skips).  This is the one test that
could pin parity to the library the reference actually calls (FACE/kmsfacedetect.cpp:805-811) rather than to the
oracle's restatement of it; OpenCV >= 3 re-implements old-format cascade evaluation, so the cascade check is
reported with the version in the assertion message."""
import os
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
cv2 = pytest.importorskip("cv2")
LEGACY = cv2.__version__.startswith("2.4")        # the line the reference was written against


def _same(ok, what):
    """2.4.x must match bit for bit; a later major version re-implements these paths, so a difference there is a finding
    to read (expected failure with the details), not a regression of this repository"""
    if ok:
        return
    msg = "%s differs from OpenCV %s" % (what, cv2.__version__)
    if LEGACY:
        pytest.fail(msg)
    pytest.xfail(msg)


@pytest.fixture(scope="module")
def ctx():
    from nubovca import capi
    c = capi.Context(0)
    yield c
    c.close()


def test_primitives_match_opencv(ctx):
    from nubovca import synth
    bgr = synth.make_bgr(640, 480, 77, "natural", [(100, 80, 200)])
    small = cv2.resize(bgr, (160, 120), interpolation=cv2.INTER_LINEAR)
    _same(np.array_equal(ctx.resize_linear(bgr, 160, 120), small), "resize")
    gray = cv2.cvtColor(small, cv2.COLOR_BGR2GRAY)
    _same(np.array_equal(ctx.bgr2gray(small), gray), "cvtColor")
    eq = cv2.equalizeHist(gray)
    _same(np.array_equal(ctx.equalize_hist(gray), eq), "equalizeHist")
    s, q = cv2.integral2(eq, sdepth=cv2.CV_32S, sqdepth=cv2.CV_64F)
    gs, gq = ctx.integral(eq)
    _same(np.array_equal(gs, s) and np.array_equal(gq, q), "integral")


def test_detect_multiscale_matches_opencv(ctx, synth_xml):
    from nubovca import synth
    casc = ctx.load_cascade_xml(synth_xml)
    fd, path = tempfile.mkstemp(suffix=".xml")
    try:
        with os.fdopen(fd, "w") as f:
            f.write(synth_xml)
        cc = cv2.CascadeClassifier(path)
        assert not cc.empty()
        gray = cv2.equalizeHist(synth.make_gray(640, 480, 3, "natural", [(100, 80, 120), (300, 200, 60)]))
        exp = np.asarray(cc.detectMultiScale(gray, scaleFactor=1.1, minNeighbors=3, flags=0, minSize=(32, 24)), np.int32).reshape(-1, 4)
        got = ctx.detect_multiscale(casc, gray, 1.1, 3, 0, (32, 24))
        _same(sorted(map(tuple, got.tolist())) == sorted(map(tuple, exp.tolist())), "detectMultiScale (%s vs %s)" % (got.tolist(), exp.tolist()))
    finally:
        os.unlink(path)
