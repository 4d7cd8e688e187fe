"""Committed golden vectors (tests/golden/, produced by tests/golden/make_golden.py from the oracle): the CPU
oracle must keep reproducing them (not gpu) and the HIP path must match them through the C ABI (gpu)."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "face_path_160x120.npz"))
XML = bytes(G["cascade_xml"]).decode()


def test_oracle_reproduces_golden():
    import orc
    c = orc.parse_cascade_xml(XML)
    gray, eq = G["gray"], G["equalized"]
    assert np.array_equal(orc.equalize_hist(gray), eq)
    assert np.array_equal(orc.resize_linear(gray, 80, 60), G["resized_80x60"])
    assert np.array_equal(orc.resize_linear(gray, 53, 41), G["resized_53x41"])
    s, q = orc.integral(eq)
    assert np.array_equal(s, G["integral_sum"]) and np.array_equal(q, G["integral_sqsum"])
    assert np.array_equal(orc.detect_raw(c, eq, 1.1, 0, (0, 0)), G["raw_sc"])
    assert np.array_equal(orc.detect_multiscale(c, eq, 1.1, 3, 0, (0, 0)), G["det_sc"])
    assert np.array_equal(orc.detect_raw(c, eq, 1.1, orc.HAAR_SCALE_IMAGE, (0, 0)), G["raw_si"])
    assert np.array_equal(orc.detect_multiscale(c, eq, 1.1, 2, orc.HAAR_SCALE_IMAGE, (0, 0)), G["det_si"])
    assert np.array_equal(orc.detect_multiscale(c, eq, 1.1, 3, orc.HAAR_FIND_BIGGEST_OBJECT, (1, 1)), G["det_big"])
    fs = orc.FaceStream(c, width_to_process=160, scale_factor_pct=10)
    tr = orc.Tracker(threshold=15, min_area=20)
    for i, f in enumerate(G["frames_bgr"]):
        b, ids = fs.process(f)
        assert np.array_equal(b, G["face_boxes_%d" % i]) and np.array_equal(ids, G["face_ids_%d" % i])
        bgra = np.concatenate([f, np.full(f.shape[:2] + (1,), 255, np.uint8)], axis=2)
        assert np.array_equal(tr.process(bgra, 100.0 + 33.0 * i), G["trk_boxes_%d" % i])


@pytest.mark.gpu
def test_hip_path_matches_golden():
    from nubovca import capi
    ctx = capi.Context(0)
    c = ctx.load_cascade_xml(XML)
    gray, eq = G["gray"], G["equalized"]
    assert np.array_equal(ctx.equalize_hist(gray), eq)
    assert np.array_equal(ctx.resize_linear(gray, 80, 60), G["resized_80x60"])
    assert np.array_equal(ctx.resize_linear(gray, 53, 41), G["resized_53x41"])
    s, q = ctx.integral(eq)
    assert np.array_equal(s, G["integral_sum"]) and np.array_equal(q, G["integral_sqsum"])
    assert np.array_equal(ctx.detect_raw(c, eq, 1.1, 0, (0, 0)), G["raw_sc"])
    assert np.array_equal(ctx.detect_multiscale(c, eq, 1.1, 3, 0, (0, 0)), G["det_sc"])
    assert np.array_equal(ctx.detect_raw(c, eq, 1.1, capi.HAAR_SCALE_IMAGE, (0, 0)), G["raw_si"])
    assert np.array_equal(ctx.detect_multiscale(c, eq, 1.1, 2, capi.HAAR_SCALE_IMAGE, (0, 0)), G["det_si"])
    assert np.array_equal(ctx.detect_multiscale(c, eq, 1.1, 3, capi.HAAR_FIND_BIGGEST_OBJECT, (1, 1)), G["det_big"])
    fs = capi.FaceStream(ctx, c, width_to_process=160, multi_scale_factor=10)
    tr = capi.Tracker(ctx, set_threshold=15, set_min_area=20)
    for i, f in enumerate(G["frames_bgr"]):
        b, ids = fs.process(f)
        assert np.array_equal(b, G["face_boxes_%d" % i]) and np.array_equal(ids, G["face_ids_%d" % i])
        bgra = np.ascontiguousarray(np.concatenate([f, np.full(f.shape[:2] + (1,), 255, np.uint8)], axis=2))
        assert np.array_equal(tr.process(bgra, 100.0 + 33.0 * i), G["trk_boxes_%d" % i])
    ctx.close()
