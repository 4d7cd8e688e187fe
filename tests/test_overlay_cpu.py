"""nvca_overlay_blend on host frames (no device, no context): against a numpy statement of the reference's loop
(FACE/kmsfacedetect.cpp:427-502), whose C-channel resize is first pinned against the oracle's 1- and 3-channel cv::resize."""
import numpy as np
import pytest

from overlay_reference import overlay_blend as ref_blend, resize_linear_cn


def test_reference_resize_equals_the_oracle_on_1_and_3_channels():
    import orc
    rng = np.random.default_rng(4)
    for (sh, sw, dh, dw) in [(40, 60, 17, 23), (33, 21, 66, 42), (50, 50, 25, 25), (7, 9, 70, 90), (64, 48, 64, 48), (12, 200, 5, 31), (31, 17, 90, 11)]:
        g = rng.integers(0, 256, (sh, sw)).astype(np.uint8)
        assert np.array_equal(resize_linear_cn(g, dw, dh), orc.resize_linear(g, dw, dh)), (sh, sw, dh, dw)
        c = rng.integers(0, 256, (sh, sw, 3)).astype(np.uint8)
        assert np.array_equal(resize_linear_cn(c, dw, dh), orc.resize_linear(c, dw, dh)), (sh, sw, dh, dw)


@pytest.mark.parametrize("cn", [1, 3, 4])
def test_overlay_host_frames(cn):
    from nubovca import capi
    rng = np.random.default_rng(10 + cn)
    img = rng.integers(0, 256, (37, 53) if cn == 1 else (37, 53, cn)).astype(np.uint8)
    if cn == 4:
        img[:, :, 3] = np.where(rng.random((37, 53)) < 0.3, 255, np.where(rng.random((37, 53)) < 0.3, 0, img[:, :, 3]))     # opaque, clear and partial pixels
    W, H = 200, 150
    cases = [
        ([(20, 30, 80, 60)], 0.0, 0.0, 1.0, 1.0),
        ([(20, 30, 80, 60), (60, 50, 90, 90)], -0.25, -0.5, 1.5, 1.25),        # overlapping boxes, image larger than the box
        ([(150, 100, 90, 80)], 0.1, 0.1, 1.0, 1.0),                           # sticks out right / bottom
        ([(-30, -20, 100, 70)], 0.0, 0.0, 1.0, 1.0),                          # sticks out left / top
        ([(10, 10, 106, 74)], 0.0, 0.0, 1.0, 1.0),                            # exactly twice the image: the 2 x 2 area shortcut does not apply upwards
        ([(10, 10, 26, 18)], 0.0, 0.0, 1.0, 1.03),                            # about half size
        ([(5, 5, 53, 37)], 0.0, 0.0, 1.0, 1.0),                               # identity scale
        ([(40, 40, 3, 2)], 0.0, 0.0, 0.4, 0.6),                               # scaled size 1 x 1
        ([(40, 40, 3, 2)], 0.0, 0.0, 0.2, 0.6),                               # scaled width 0: skipped
        ([(40, 40, 60, 60)], 0.3, 0.7, 0.0, 1.0),                             # width_percent 0: nothing drawn
    ]
    for boxes, ox, oy, wp, hp in cases:
        frame = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
        exp = ref_blend(frame.copy(), boxes, img, ox, oy, wp, hp)
        got = frame.copy()
        capi.overlay_blend(None, got, boxes, img, ox, oy, wp, hp)
        assert np.array_equal(got, exp), (cn, boxes, ox, oy, wp, hp, int((got != exp).sum()))
    half = rng.integers(0, 256, (40, 60, cn) if cn > 1 else (40, 60)).astype(np.uint8)          # exactly half size: the area shortcut
    frame = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
    exp = ref_blend(frame.copy(), [(30, 20, 30, 20)], half)
    got = frame.copy()
    capi.overlay_blend(None, got, [(30, 20, 30, 20)], half)
    assert np.array_equal(got, exp)


def test_overlay_arguments_are_checked():
    from nubovca import capi
    frame = np.zeros((20, 20, 3), np.uint8)
    with pytest.raises(capi.NvcaError):
        capi.overlay_blend(None, frame, [(0, 0, 5, 5)], np.zeros((4, 4, 2), np.uint8))          # 2 channels
    with pytest.raises(capi.NvcaError):
        capi.overlay_blend(None, frame, [(0, 0, 5, 5)], np.zeros((4, 4, 3), np.uint8), width=1e9)
    capi.overlay_blend(None, frame, [], np.zeros((4, 4, 3), np.uint8))
