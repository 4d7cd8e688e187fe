"""Pins the CPU oracle's imgproc restatement against closed-form known answers
(SURVEY.md 8c / Appendix A.1-A.4).  The reference holds no fixtures for this
path, so these hand-derived cases are the only anchors ("parity unpinned")."""
import numpy as np
import pytest
import orc


def test_gray_primaries():
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0]]], np.uint8)
    assert orc.bgr2gray(px).tolist() == [[29, 150, 76, 255, 0]]


def test_gray_formula_random_and_bgra():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(37, 53, 4), dtype=np.uint8)
    exp = ((img[..., 0].astype(np.int64) * 1868 + img[..., 1].astype(np.int64) * 9617
            + img[..., 2].astype(np.int64) * 4899 + 8192) >> 14).astype(np.uint8)
    assert np.array_equal(orc.bgr2gray(img), exp)
    assert np.array_equal(orc.bgr2gray(img[..., :3]), exp)


def test_resize_identity_and_constant():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(31, 45), dtype=np.uint8)
    assert np.array_equal(orc.resize_linear(img, 45, 31), img)
    c = np.full((40, 60, 3), 77, np.uint8)
    assert np.all(orc.resize_linear(c, 13, 9) == 77)


def test_resize_half_is_area_average():
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(20, 32), dtype=np.uint8).astype(np.int32)
    exp = (img[0::2, 0::2] + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2
    assert np.array_equal(orc.resize_linear(img.astype(np.uint8), 16, 10), exp.astype(np.uint8))


def test_resize_integer_ratio_samples_two_taps():
    # scale 4: fx = (dx+0.5)*4-0.5 = 4dx+1.5 -> sx = 4dx+1, fx=.5 -> mean of taps 4dx+1, 4dx+2
    src = np.zeros((8, 16), np.uint8)
    src[:, 1::4] = 100
    src[:, 2::4] = 200
    out = orc.resize_linear(src, 4, 2)
    assert np.all(out == 150)


def test_resize_upscale_edges_clamp():
    src = np.array([[0, 100]], np.uint8)
    out = orc.resize_linear(src, 4, 1)
    # fx: -0.25 -> clamp 0 ; 0.25 ; 0.75 ; 1.25 -> clamp to last
    assert out.tolist() == [[0, 25, 75, 100]]


def test_equalize_two_levels_and_constant():
    img = np.zeros((10, 10), np.uint8)
    img[:, 5:] = 200
    out = orc.equalize_hist(img)
    assert set(np.unique(out)) == {0, 255}
    const = np.full((7, 9), 42, np.uint8)
    assert np.all(orc.equalize_hist(const) == 42)


def test_equalize_matches_numpy_formula():
    rng = np.random.default_rng(3)
    img = rng.integers(10, 200, size=(64, 48), dtype=np.uint8)
    hist = np.bincount(img.ravel(), minlength=256)
    i = int(np.nonzero(hist)[0][0])
    scale = np.float32(255.0) / np.float32(img.size - hist[i])
    lut = np.zeros(256, np.uint8)
    s = 0
    for j in range(i + 1, 256):
        s += int(hist[j])
        lut[j] = np.clip(np.rint(np.float32(s) * scale), 0, 255)
    assert np.array_equal(orc.equalize_hist(img), lut[img])


def test_integral_ones_and_random():
    s, q = orc.integral(np.ones((5, 7), np.uint8))
    yy, xx = np.mgrid[0:6, 0:8]
    assert np.array_equal(s, yy * xx)
    assert np.array_equal(q, (yy * xx).astype(np.float64))
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, size=(33, 41), dtype=np.uint8)
    s, q = orc.integral(img)
    assert np.array_equal(s[1:, 1:], img.astype(np.int64).cumsum(0).cumsum(1))
    assert np.array_equal(q[1:, 1:], (img.astype(np.int64) ** 2).cumsum(0).cumsum(1).astype(np.float64))
    assert not s[0].any() and not s[:, 0].any()


def test_flip():
    img = np.arange(12, dtype=np.uint8).reshape(3, 4)
    assert np.array_equal(orc.flip_h(img), img[:, ::-1])


def test_gray_of_grey_pixels_is_identity():
    """the fixed-point weights add up to 1 << 14: (v, v, v) -> v for every v"""
    import orc
    v = np.arange(256, dtype=np.uint8)
    img = np.repeat(v[None, :, None], 3, axis=2).copy()
    assert np.array_equal(orc.bgr2gray(img)[0], v)


def test_integral_closed_forms():
    """horizontal ramp p(x, y) = x: sum[y][x] = y * x (x - 1) / 2, sqsum[y][x] = y * (x - 1) x (2 x - 1) / 6"""
    import orc
    w, h = 200, 37
    img = np.tile(np.arange(w, dtype=np.uint8), (h, 1))
    s, q = orc.integral(img)
    X, Y = np.meshgrid(np.arange(w + 1, dtype=np.int64), np.arange(h + 1, dtype=np.int64))
    assert np.array_equal(s.astype(np.int64), Y * X * (X - 1) // 2)
    assert np.array_equal(q, (Y * (X - 1) * X * (2 * X - 1) // 6).astype(np.float64))


def test_equalize_ramp_with_equal_bins():
    """256 levels, n pixels each: lut[j] = cvRound(j * n * 255 / (256 n - n)) = j (the scale is exactly 1 / n)"""
    import orc
    n = 5
    img = np.repeat(np.arange(256, dtype=np.uint8), n).reshape(16, 16 * n)
    assert np.array_equal(orc.equalize_hist(img), img)


def test_resize_constant_rows_stay_constant():
    """bilinear weights of a destination pixel add up to 2048 * 2048: a constant image stays constant at any ratio"""
    import orc
    for (sw, sh, dw, dh) in [(97, 61, 41, 29), (640, 480, 213, 160), (33, 17, 100, 50)]:
        img = np.full((sh, sw), 173, np.uint8)
        assert np.all(orc.resize_linear(img, dw, dh) == 173)


# ------------------------------------------------------------------ tilted integral (cv::integral's third output)
def _tilted_by_definition(img):
    """tilted(X,Y) = sum of image(x,y) over y < Y, abs(x - X + 1) <= Y - y - 1  (OpenCV 2.4 imgproc documentation)"""
    h, w = img.shape
    T = np.zeros((h + 1, w + 1), np.int64)
    for Y in range(h + 1):
        for X in range(w + 1):
            for y in range(Y):
                lo, hi = max(0, X - 1 - (Y - y - 1)), min(w - 1, X - 1 + (Y - y - 1))
                if hi >= lo:
                    T[Y, X] += int(img[y, lo:hi + 1].sum())
    return T.astype(np.int32)


@pytest.mark.parametrize("h,w", [(1, 1), (1, 6), (6, 1), (2, 2), (7, 5), (6, 11), (13, 13), (3, 20), (20, 3)])
def test_tilted_integral_matches_the_published_definition(h, w):
    img = np.random.default_rng(h * 31 + w).integers(0, 256, size=(h, w)).astype(np.uint8)
    assert np.array_equal(orc.integral_tilted(img), _tilted_by_definition(img))


def test_tilted_integral_known_answers():
    # first row and the apex rule: tilted(X, 1) = image(X-1, 0); tilted(X, 2) adds the three pixels above the apex
    img = np.arange(1, 13, dtype=np.uint8).reshape(3, 4)          # rows [1 2 3 4], [5 6 7 8], [9 10 11 12]
    T = orc.integral_tilted(img)
    assert T[0].tolist() == [0, 0, 0, 0, 0]
    assert T[1].tolist() == [0, 1, 2, 3, 4]
    assert T[2].tolist() == [1, 5 + 1 + 2, 6 + 1 + 2 + 3, 7 + 2 + 3 + 4, 8 + 3 + 4]      # column 0: only pixel (0,0) lies under apex (-1,1)
    # a tilted rectangle (x, y, w, h) covers exactly 2*w*h pixels: on a constant image its four-corner sum is 2*w*h*v
    flat = np.full((40, 40), 7, np.uint8)
    T = orc.integral_tilted(flat).astype(np.int64)
    for (x, y, w, h) in [(10, 3, 4, 3), (12, 2, 6, 2), (20, 5, 2, 7)]:
        assert T[y, x] - T[y + h, x - h] - T[y + w, x + w] + T[y + w + h, x + w - h] == 2 * w * h * 7
    # whole image under a far apex: the last row's widest triangle
    ones = np.ones((5, 9), np.uint8)
    assert orc.integral_tilted(ones)[5, 5] == 1 + 3 + 5 + 7 + 9
