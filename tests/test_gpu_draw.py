"""view-* outlines: nvca_draw_shapes on host frames (what the GStreamer shim calls) and on device frames
(a viewed stream that stays in HBM) against the reference of tests/draw_reference.py, written from the rule include/nubovca.h states: the rectangle
of cvRectangle(..., thickness 3) as the union of its four 3-pixel edge bands plus the 4-neighbourhood of each vertex
(FACE/kmsfacedetect.cpp:836-845), the circle of circle(..., thickness 4) as the ring of distances [r - 2, r + 2]
(EYE/kmseyedetect.cpp:1075-1092).  Not pinned against OpenCV's rasteriser (no OpenCV offline): the three implementations
(numpy here, the library's host loop, the kernel) agree pixel for pixel."""
import numpy as np
import pytest

from draw_reference import _ref, _shapes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from nubovca import capi
    c = capi.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("C,W,H,n", [(3, 160, 120, 7), (4, 333, 251, 12), (3, 1920, 1080, 40), (4, 64, 48, 1), (3, 40, 30, 0)])
def test_host_and_device_outlines(ctx, C, W, H, n):
    import torch
    from nubovca import capi
    rng = np.random.default_rng(W * 7 + n)
    base = rng.integers(0, 256, size=(H, W, C), dtype=np.uint8)
    shapes = _shapes(rng, W, H, n) + ([(0, 5, 5, 0, 0, (1, 2, 3, 4)), (0, W - 2, H - 2, 10, 10, (9, 8, 7, 6)), (1, W // 2, H // 2, 0, 0, (5, 5, 5, 5)),
                                       (1, W // 2, H // 2, 1, 0, (6, 6, 6, 6))] if n else [])
    exp = _ref(base.copy(), shapes)
    host = base.copy()
    ctx.draw_shapes(host, C, shapes)
    assert np.array_equal(host, exp)
    dev = torch.from_numpy(base.copy()).cuda()
    torch.cuda.synchronize()
    ctx.draw_shapes(capi.make_frame(dev.data_ptr(), W, H, W * C, capi.MEM_DEVICE), C, shapes)
    ctx.synchronize()
    assert np.array_equal(dev.cpu().numpy(), exp)


def test_padded_rows_and_bad_arguments(ctx):
    from nubovca import capi
    img = np.zeros((50, 80, 4), np.uint8)
    view = img[:, :61, :3]                               # a 61-pixel BGR image... not contiguous: hand the stride over
    fr = capi.Frame(img.ctypes.data, 61, 50, img.strides[0], capi.MEM_HOST, 0)
    ctx.draw_shapes(fr, 4, [(0, 10, 10, 100, 20, (1, 2, 3, 4))])
    exp = _ref(np.zeros((50, 61, 4), np.uint8), [(0, 10, 10, 100, 20, (1, 2, 3, 4))])
    assert np.array_equal(img[:, :61], exp) and not img[:, 61:].any() and view.shape == (50, 61, 3)
    with pytest.raises(capi.NvcaError):
        ctx.draw_shapes(np.zeros((8, 8, 3), np.uint8), 3, [(7, 0, 0, 1, 1, (0, 0, 0, 0))])
    with pytest.raises(capi.NvcaError):
        ctx.draw_shapes(np.zeros((8, 8, 3), np.uint8), 2, [])


@pytest.mark.parametrize("cn", [1, 3, 4])
def test_overlay_on_device_frames(ctx, cn):
    """nvca_overlay_blend on a frame that lives in HBM (k_overlay, one launch per box): the pixels of the host loop / of the
    numpy statement of FACE/kmsfacedetect.cpp:427-502, bit for bit (the blend is f64 arithmetic, truncated)"""
    import torch
    from nubovca import capi
    from overlay_reference import overlay_blend as ref_blend
    rng = np.random.default_rng(20 + cn)
    img = rng.integers(0, 256, (45, 70) if cn == 1 else (45, 70, cn)).astype(np.uint8)
    W, H = 640, 360
    for boxes, ox, oy, wp, hp in [([(100, 60, 200, 160), (220, 120, 180, 200)], -0.1, -0.3, 1.2, 1.4), ([(560, 300, 160, 120), (-40, -30, 120, 90)], 0.0, 0.0, 1.0, 1.0),
                                  ([(10, 10, 140, 90)], 0.0, 0.0, 1.0, 1.0), ([(300, 100, 35, 22)], 0.0, 0.0, 1.0, 1.03)]:
        frame = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
        exp = ref_blend(frame.copy(), boxes, img, ox, oy, wp, hp)
        host = frame.copy()
        capi.overlay_blend(None, host, boxes, img, ox, oy, wp, hp)
        dev = torch.from_numpy(frame.copy()).cuda()
        torch.cuda.synchronize()
        capi.overlay_blend(ctx, capi.make_frame(dev.data_ptr(), W, H, W * 3, capi.MEM_DEVICE), boxes, img, ox, oy, wp, hp)
        got = dev.cpu().numpy()
        assert np.array_equal(host, exp) and np.array_equal(got, exp), (cn, boxes, int((got != exp).sum()))
