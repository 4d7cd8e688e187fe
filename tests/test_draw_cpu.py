"""nvca_draw_shapes on host frames needs neither a device nor a context: the library's host rasteriser (what the GStreamer
shim calls for view-faces / view-eyes / ...) against the reference in tests/draw_reference.py, on the CPU."""
import numpy as np
import pytest

from draw_reference import _ref, _shapes


@pytest.mark.parametrize("C,W,H,n", [(3, 160, 120, 9), (4, 333, 251, 14), (3, 64, 48, 1), (4, 40, 30, 0)])
def test_host_outlines_without_a_device(C, W, H, n):
    from nubovca import capi
    rng = np.random.default_rng(W + 31 * n)
    base = rng.integers(0, 256, size=(H, W, C), dtype=np.uint8)
    shapes = _shapes(rng, W, H, n) + ([(0, 5, 5, 0, 0, (1, 2, 3, 4)), (0, W - 2, H - 2, 10, 10, (9, 8, 7, 6)), (1, W // 2, H // 2, 0, 0, (5, 5, 5, 5)),
                                       (0, 30, 20, -25, -15, (7, 7, 7, 7))] if n else [])
    img = base.copy()
    capi.draw_shapes_host(img, C, shapes)
    assert np.array_equal(img, _ref(base.copy(), shapes))


def test_host_outline_arguments():
    from nubovca import capi
    with pytest.raises(capi.NvcaError):
        capi.draw_shapes_host(np.zeros((8, 8, 3), np.uint8), 3, [(5, 0, 0, 1, 1, (0, 0, 0, 0))])      # unknown kind
    with pytest.raises(capi.NvcaError):
        capi.draw_shapes_host(np.zeros((8, 8, 3), np.uint8), 2, [])                                      # channels
