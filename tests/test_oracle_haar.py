"""Pins the CPU oracle's Haar restatement against
hand-derived known answers and against the counts SURVEY.md 8a derives
independently from A.5's loop."""
import numpy as np
import orc
from nubovca import synth


def _count(W, H, sf, ms):
    fs = orc.scale_grid(20, 20, W, H, sf, ms)
    tot = 0
    for f in fs:
        ystep = max(2.0, f)
        ww = int(np.rint(20 * f))
        tot += int(np.rint((W - ww) / ystep)) * int(np.rint((H - ww) / ystep))
    return len(fs), tot


def test_scale_grid_matches_survey_counts():
    # SURVEY.md 8a row a6
    assert _count(1920, 1080, 1.1, (96, 54)) == (25, 355162)
    assert _count(1280, 720, 1.1, (64, 36)) == (25, 336037)
    assert _count(160, 90, 1.25, (8, 4)) == (7, 9279)
    assert _count(160, 120, 1.25, (8, 6)) == (8, 14227)


def test_group_rectangles_known_answers():
    r = [[10, 10, 20, 20]] * 4
    out, w = orc.group_rectangles(r, 3)
    assert out.tolist() == [[10, 10, 20, 20]] and w.tolist() == [4]
    out, _ = orc.group_rectangles([[10, 10, 20, 20]] * 3, 3)      # n must be > threshold
    assert len(out) == 0
    out, w = orc.group_rectangles([[1, 2, 3, 4], [50, 60, 7, 8]], 0)   # threshold 0: untouched
    assert out.tolist() == [[1, 2, 3, 4], [50, 60, 7, 8]] and w.tolist() == [1, 1]


def test_group_rectangles_average_rounds_half_even_and_order():
    # two clusters; class order follows first member in input order
    a = [[100, 100, 40, 40], [101, 100, 40, 40], [100, 101, 40, 40], [101, 101, 41, 41]]
    b = [[10, 10, 20, 20], [11, 10, 20, 20], [10, 10, 20, 20], [11, 11, 20, 20], [10, 11, 20, 20]]
    rects = [a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], b[4]]
    out, w = orc.group_rectangles(rects, 3)
    # a: x sum 402 -> 100.5 -> 100 (half-even); y 402 -> 100; w 161 -> 40.25 -> 40
    assert out.tolist() == [[100, 100, 40, 40], [10, 10, 20, 20]]
    assert w.tolist() == [4, 5]


def test_group_rectangles_drops_small_inside_big():
    big = [[100, 100, 100, 100]] * 6
    small = [[130, 130, 30, 30]] * 4
    out, w = orc.group_rectangles(big + small, 3)
    assert out.tolist() == [[100, 100, 100, 100]]
    # equal support: n2 > max(3, n1) fails -> both kept
    out, w = orc.group_rectangles([[100, 100, 100, 100]] * 4 + small, 3)
    assert len(out) == 2


def _one_stump_cascade(thr, left, right, stage_thr, rects=((2, 2, 8, 4, -1.0), (6, 2, 4, 4, 2.0))):
    c = dict(name="t", size=(12, 12),
             stages=[dict(features=[list(rects)], thresholds=[thr], left=[left], right=[right],
                          stage_threshold=stage_thr)])
    return orc.parse_cascade_xml(synth.cascade_to_xml(c))


def test_single_stump_hand_computed():
    """12x12 window, one x2 edge stump over (2,2,8,4): left half dark, right half bright.

    equRect = (1,1,10,10), inv_area = 1/100.  Image: columns < 6 are 0, columns >= 6 are 200.
    rect0 (2,2,8,4) sum = 4 cols*4 rows*200 = 3200, rect1 (6,2,4,4) sum = 3200.
    w1 = 2/100 = .02 ; w0 = -(w1*16)/32 = -.01 ; feature = 3200*-.01 + 3200*.02 = 32.
    mean = (5 cols*10 rows*200)/100 = 100 ; sq mean = 5*10*40000/100 = 20000 ; var = 10000 ; std 100.
    normalised value .32 : passes iff threshold <= .32.
    """
    img = np.zeros((23, 23), np.uint8)      # 12x12 window at (0,0) sees exactly the pattern above
    img[:, 6:] = 200
    for thr, expect in ((0.31, True), (0.33, False)):
        c = _one_stump_cascade(thr, -1.0, 1.0, 0.5)
        raw = orc.detect_raw(c, img, 1.1, 0)
        assert ([0, 0, 12, 12] in raw.tolist()) == expect, (thr, raw)


def test_window_touching_border_is_rejected():
    """runCascade's bound is x + w >= sum.cols -> -1: with a 12x12 image and window the only
    position is (0,0); endX = round((12-12)/2) = 0 -> nothing is even scanned at 12x12,
    while a 23x23 image scans exactly the loop-limited grid."""
    c = _one_stump_cascade(-1e9, 1.0, 1.0, 0.5)      # accepts everything
    img = np.zeros((23, 23), np.uint8)
    raw = orc.detect_raw(c, img, 1.1, 0)
    # factor loop: f*12 < 13 -> only f = 1 ; endX = endY = round(11/2) = 6 (half-even of 5.5)
    # stage passes -> result 1 -> ixstep 1 : all 36 positions, x = 0,2,..,10
    assert len(raw) == 36
    assert raw[:, 0].max() == 10 and raw[:, 2].max() == 12


def test_adaptive_xstep_skips_after_stage0_reject():
    c = _one_stump_cascade(1e9, -1.0, 1.0, 0.5)       # rejects every window of non-zero variance
    img = np.random.default_rng(0).integers(0, 256, size=(23, 23), dtype=np.uint8)
    raw, st = orc.detect_raw(c, img, 1.1, 0, return_stats=True)
    assert len(raw) == 0
    assert st.windows == 6 * 3                          # ix = 0,2,4 per row


def test_synthetic_faces_found(orc_cascade):
    faces = [(100, 80, 120), (300, 200, 60), (420, 60, 97)]
    g = orc.equalize_hist(synth.make_gray(640, 480, 1, "natural", faces))
    boxes = orc.detect_multiscale(orc_cascade, g, 1.1, 3, 0, (32, 24))
    assert len(boxes) == 3
    for (x, y, s) in faces:
        d = np.abs(boxes - np.array([x, y, s, s])).max(axis=1).min()
        assert d <= 0.15 * s


def test_policies_agree_on_3rect_stage_and_can_differ_on_pairs(orc_small):
    g = orc.equalize_hist(synth.make_gray(320, 240, 5, "natural", [(60, 40, 100)]))
    a = orc.detect_raw(orc_small, g, 1.1, 0, (30, 30), policy=orc.SUM_F32PAIR)
    b = orc.detect_raw(orc_small, g, 1.1, 0, (30, 30), policy=orc.SUM_F64)
    assert len(a) > 0 and len(b) > 0      # both run; equality is not required by the spec


def test_scale_image_and_biggest_variants_run(orc_cascade):
    g = orc.equalize_hist(synth.make_gray(200, 160, 9, "natural", [(40, 30, 80)]))
    want = np.array([40, 30, 80, 80])
    si = orc.detect_multiscale(orc_cascade, g, 1.1, 2, orc.HAAR_SCALE_IMAGE, (20, 20))
    assert len(si) >= 1 and np.abs(si - want).max(axis=1).min() <= 10
    big = orc.detect_multiscale(orc_cascade, g, 1.1, 3, orc.HAAR_FIND_BIGGEST_OBJECT, (1, 1))
    assert len(big) == 1 and np.abs(big[0] - want).max() <= 10


def test_scale_image_raw_grid_hand_checked():
    c = _one_stump_cascade(-1e9, 1.0, 1.0, 0.5)       # accepts everything, 12x12 window
    img = np.zeros((16, 20), np.uint8)
    raw = orc.detect_raw(c, img, 1.5, orc.HAAR_SCALE_IMAGE)
    # f=1: sz 20x16, y in [0,4) step 2, x in [0,8) step 2 -> 2*4 = 8 hits of 12x12
    # f=1.5: win 18, sz = (round(13.33), round(10.67)) = 13x11 -> sz1 = 2x0 -> break
    assert len(raw) == 8 and set(raw[:, 2]) == {12}


# ------------------------------------------------------------------ tilted features and tree weak classifiers
def _tilted_rect_sum(img, x, y, w, h):
    """pixel sum of OpenCV's tilted rectangle from the definition of the tilted integral (independent of orc_haar.c)"""
    T = orc.integral_tilted(img).astype(np.int64)
    return int(T[y, x] - T[y + h, x - h] - T[y + w, x + w] + T[y + w + h, x + w - h])


def test_tilted_stump_hand_computed():
    """12x12 window, one tilted stump: rect0 = (6,1,4,2) tilted weight -1, rect1 = (6,1,2,2) tilted weight 2.
    equRect (1,1,10,10), inv_area 1/100, tilted correction 0.5: w1 = 2 * 0.005 = .01, w0 = -(w1 * 2*2) / (4*2) = -.005
    (areas as cvSetImages computes them: width*height of the unrotated rect).  On an image that is 0 except a 200-valued
    block covering rect1's pixels the feature is s0*w0 + s1*w1 with s0 = s1 = (pixels of rect1) * 200."""
    img = np.zeros((23, 23), np.uint8)
    # rect1 (6,1,2,2) tilted covers 8 pixels; paint them through the one-hot probe so the test does not re-derive geometry
    pix = []
    for py in range(12):
        for px in range(12):
            probe = np.zeros((23, 23), np.uint8); probe[py, px] = 1
            if _tilted_rect_sum(probe, 6, 1, 2, 2):
                pix.append((py, px))
    assert len(pix) == 8
    for (py, px) in pix:
        img[py, px] = 200
    s1 = _tilted_rect_sum(img, 6, 1, 2, 2); s0 = _tilted_rect_sum(img, 6, 1, 4, 2)
    assert s1 == 1600 and s0 == 1600                      # rect1's pixels are inside rect0
    feat = np.float32(s0) * np.float32(-0.005) + np.float32(s1) * np.float32(0.01)       # = 8
    inner = img[1:11, 1:11].astype(np.float64)
    std = np.sqrt((inner ** 2).sum() / 100 - (inner.sum() / 100) ** 2)
    val = float(feat) / std
    tilted_stage = lambda thr: dict(name="t", size=(12, 12), stages=[dict(
        features=[[(6, 1, 4, 2, -1.0), (6, 1, 2, 2, 2.0)]], tilted=[1], thresholds=[thr], left=[-1.0], right=[1.0], stage_threshold=0.5)])
    for thr, expect in ((val - 0.01, True), (val + 0.01, False)):
        c = orc.parse_cascade_xml(synth.cascade_to_xml(tilted_stage(thr)))
        assert int(c.tilted.sum()) == 1
        raw = orc.detect_raw(c, img, 1.1, 0)
        assert ([0, 0, 12, 12] in raw.tolist()) == expect, (thr, val, raw)


def test_two_node_tree_hand_computed():
    """one weak classifier with two nodes on the image of test_single_stump_hand_computed (x2 edge feature, normalised value
    .32 at window (0,0)): root sends value < t0 to the leaf -1 and value >= t0 to node 1; node 1 (same feature) votes -1
    below t1 and +1 from t1 on.  The window passes the stage (threshold .5) iff value >= t0 and value >= t1."""
    img = np.zeros((23, 23), np.uint8)
    img[:, 6:] = 200
    feat = [(2, 2, 8, 4, -1.0), (6, 2, 4, 4, 2.0)]

    def casc(t0, t1):
        tree = [dict(feature=feat, tilted=0, threshold=t0, left=("val", -1.0), right=("node", 1)),
                dict(feature=feat, tilted=0, threshold=t1, left=("val", -1.0), right=("val", 1.0))]
        return orc.parse_cascade_xml(synth.cascade_to_xml(dict(name="t", size=(12, 12), stages=[dict(trees=[tree], stage_threshold=0.5)])))
    for t0, t1, expect in ((0.31, 0.31, True), (0.33, 0.31, False), (0.31, 0.33, False), (0.1, 0.2, True)):
        c = casc(t0, t1)
        assert c.cls_nnodes.tolist() == [2] and c.left.tolist() == [0, -1] and c.right.tolist() == [1, -2]
        raw = orc.detect_raw(c, img, 1.1, 0)
        assert ([0, 0, 12, 12] in raw.tolist()) == expect, (t0, t1, raw)


def test_generic_cascade_all_scan_variants_run():
    """tilted + tree cascade (synth.make_generic_cascade) through the three detectMultiScale variants; the tree walk and the
    tilted plane are exercised on thousands of windows, and SCALE_IMAGE (tilted integral per pyramid level) agrees with the
    scale-cascade scan on where the strongest cluster is"""
    oc = orc.parse_cascade_xml(synth.generic_cascade_xml(seed=3))
    assert int(oc.tilted.sum()) > 10 and int((oc.cls_nnodes > 1).sum()) > 10
    img = orc.equalize_hist(synth.make_gray(200, 150, 5, "natural"))
    raw, st = orc.detect_raw(oc, img, 1.2, 0, return_stats=True)
    assert len(raw) > 5 and st.stumps > 10 * st.windows
    assert len(orc.detect_raw(oc, img, 1.2, orc.HAAR_SCALE_IMAGE)) > 5
    assert len(orc.detect_multiscale(oc, img, 1.2, 2, orc.HAAR_FIND_BIGGEST_OBJECT, (1, 1))) == 1


# ------------------------------------------------------------------ hand-written old-format files (tests/golden/oldformat_*.xml)
def _golden(name):
    import os
    return open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)).read()


def test_oldformat_stump_file_known_answers():
    """a file in cvSave's exact layout (comment block ahead of the root, closing tags glued to the last child, "-1." / "3."
    weights, three-digit exponents) -- not one written by synth.cascade_to_xml: what the loader must read out of it"""
    import orc
    c = orc.parse_cascade_xml(_golden("oldformat_stumps_24x24.xml"))
    assert (c.ow, c.oh) == (24, 24)
    assert list(c.stage_ncls) == [2, 3, 2] and list(c.cls_nnodes) == [1] * 7
    assert np.allclose(c.stage_thr, np.float32([-0.6000000238418579, -0.8125, -0.5625]), rtol=0, atol=0)
    assert np.array_equal(c.rects[0], [[4, 6, 16, 6], [4, 9, 16, 3], [0, 0, 0, 0]]) and list(c.rweights[0]) == [-1.0, 2.0, 0.0]
    assert np.array_equal(c.rects[2], [[2, 4, 20, 12], [2, 4, 10, 6], [12, 10, 10, 6]]) and list(c.rweights[2]) == [-1.0, 2.0, 2.0]
    assert list(c.rweights[1]) == [-1.0, 3.0, 0.0]
    assert c.node_thr[0] == np.float32(2.5e-3) and c.node_thr[1] == np.float32(-1.25e-2) and c.node_thr[5] == np.float32(1.0000000474974513e-3)
    assert list(c.left) == [0] * 7 and list(c.right) == [-1] * 7                 # leaves: alpha[0], alpha[1] of each stump
    assert list(c.alpha[:4]) == [-0.75, 0.625, 0.5, -0.375] and len(c.alpha) == 14
    assert not c.tilted.any()


def test_oldformat_tree_file_known_answers():
    """<left_node> / <right_node> trees and <tilted>1</tilted>: child indices and leaf numbering as icvReadHaarClassifier
    assigns them (a leaf becomes alpha[last++] with child index -last)"""
    import orc
    c = orc.parse_cascade_xml(_golden("oldformat_trees_tilted_20x20.xml"))
    assert (c.ow, c.oh) == (20, 20) and list(c.stage_ncls) == [2, 2] and list(c.cls_nnodes) == [2, 1, 3, 2]
    assert list(c.tilted) == [0, 1, 1, 0, 0, 1, 0, 0]
    assert list(c.left) == [1, -1, 0, 1, 0, -2, 0, -1] and list(c.right) == [0, -2, -1, 2, -1, -3, 1, -2]
    assert list(c.alpha) == [0.5, -0.625, 0.125, 0.375, -0.375, -0.4375, 0.3125, 0.5625, -0.1875, 0.25, -0.3125, 0.4375]
    assert np.array_equal(c.rects[1][:2], [[8, 2, 6, 3], [8, 2, 3, 3]])
    # and it evaluates: every scan variant runs, the lenient two-stage cascade lets windows through
    from nubovca import synth
    g = orc.equalize_hist(synth.make_gray(120, 90, 5, "natural"))
    assert len(orc.detect_raw(c, g, 1.2, 0, (0, 0))) > 0
    assert len(orc.detect_raw(c, g, 1.2, orc.HAAR_SCALE_IMAGE, (0, 0))) > 0
