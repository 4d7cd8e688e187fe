"""The oracle against libraries it shares no author with (fixtures written by tests/golden/make_crosscheck.py in the build
container: scipy.ndimage.label, skimage.transform.integral_image / integrate).  Two places where oracle and product could
share a misreading without any parity test noticing: the order and extent of motion components, and the corner arithmetic
of rectangle sums.  (Parity with OpenCV itself stays unpinned: DESIGN.md 2.)"""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_segment_motion_matches_scipy_label_on_plateaus():
    """every moving region holds ONE timestamp (a plateau): cvSegmentMotion's flood fill with tolerance then finds exactly the
    4-connected regions, in raster order of their first pixel -- which is what scipy.ndimage.label + find_objects return"""
    import orc
    z = np.load(os.path.join(GOLD, "crosscheck_ccl.npz"))
    comps = 0
    for i in range(int(z["n"])):
        mhi, ts, exp = z["mhi_%d" % i], float(z["ts_%d" % i]), z["boxes_%d" % i]
        got = orc.segment_motion(mhi.copy(), ts, 32.0)
        assert np.array_equal(np.asarray(got).reshape(-1, 4), exp), (i, len(got), len(exp))
        comps += len(exp)
    assert comps > 200


def test_integral_and_rect_sums_match_skimage():
    import orc
    z = np.load(os.path.join(GOLD, "crosscheck_rects.npz"))
    for i in range(int(z["n"])):
        img, ii = z["img_%d" % i], z["ii_%d" % i]
        s, _sq = orc.integral(img)
        s = np.asarray(s, np.int64)
        assert s.shape == (img.shape[0] + 1, img.shape[1] + 1) and not s[0].any() and not s[:, 0].any()      # cv::integral: a zero row and column ahead
        assert np.array_equal(s[1:, 1:], ii)
        for (x, y, w, h), e in zip(z["rects_%d" % i], z["sums_%d" % i]):
            # the corners every Haar rectangle of the oracle / the kernels reads (cvSetImagesForHaarClassifierCascade: p0 - p1 - p2 + p3)
            assert s[y + h, x + w] - s[y, x + w] - s[y + h, x] + s[y, x] == e, (i, x, y, w, h)
