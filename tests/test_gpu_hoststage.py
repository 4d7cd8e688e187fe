"""Small host images go in and out through page-locked memory of the context (csrc/api.cpp stage_2d / unstage_2d: 8 MB handed out
front to back, the device drained when it is used up).  More than that in one context -- odd widths, images of every size up
to the 2 MB limit and one beyond it (the runtime's own 2-D copy) -- every result against the oracle."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def ctx():
    from nubovca import capi
    c = capi.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
def test_more_small_host_images_than_the_staging_memory_holds(ctx):
    import orc
    rng = np.random.default_rng(21)
    moved = 0
    shapes = [(97, 83), (25, 25), (641, 479), (1280, 720), (333, 1), (1, 257), (1919, 1079), (2048, 1100)]       # the last one is beyond the limit
    for it in range(48):
        w, h = shapes[it % len(shapes)]
        img = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
        assert np.array_equal(ctx.equalize_hist(img), orc.equalize_hist(img)), (it, w, h)
        moved += 2 * ((w + 63) // 64 * 64) * h
        if it % 6 == 0:
            s, q = ctx.integral(img)
            es, eq = orc.integral(img)
            assert np.array_equal(s, es) and np.array_equal(q, eq), (it, w, h)
    assert moved > 3 * (8 << 20)            # the staging memory was used up and started over several times
