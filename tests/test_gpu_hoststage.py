"""Caller host memory reaches the HIP runtime as a raw pointer only inside a range the caller registered (nvca_host_register);
everything else crosses through page-locked slots of the context's own (csrc/api.cpp caller_h2d / caller_h2d_rows / caller_d2h_rows:
12 slots of 8 MB, each waited for before it is used again).  More than the ring holds in one context -- odd widths, images
from one pixel row to several slots -- and memory that WAS registered and is pageable again (the history of the round-3 memory
access fault); every result against the oracle."""
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
    assert moved > 3 * (8 << 20)            # several turns of the ring


@pytest.mark.gpu
def test_released_ranges_are_pageable_memory_again(ctx, synth_xml):
    """the order of calls that preceded the round-3 fault: host frames page-locked, used by a batched call, released; then small and
    large images out of the SAME heap memory through the pageable path (detectMultiScale on a 97 x 83 image among them).  Registering
    a range twice, or releasing a pointer that was never registered, is an argument error, not a call into the runtime."""
    import orc
    from nubovca import capi, synth
    casc = ctx.load_cascade_xml(synth_xml)
    oc = orc.parse_cascade_xml(synth_xml)
    W, H, N = 400, 300, 10
    pool = np.empty((N, H, W, 3), np.uint8)                      # one block: the frames, later the small images, live in the same pages
    for i in range(N):
        pool[i] = synth.make_bgr(W, H, 1900 + i, "natural", [(20 + 10 * i, 30, 120)])
    for rep in range(3):
        for i in range(N):
            ctx.host_register(pool[i])
        with pytest.raises(capi.NvcaError):
            ctx.host_register(pool[0])                           # already registered
        streams = [capi.FaceStream(ctx, casc, width_to_process=W, multi_scale_factor=10) for _ in range(N)]
        tk = ctx.face_batch_submit(streams, [capi.make_frame(pool[i]) for i in range(N)])
        for i in range(N):
            ctx.host_unregister(pool[i])                         # with the batch still in flight: every stream that carried a copy is drained first
        res = ctx.face_batch_collect(tk)
        for i in range(N):
            eb, eid = orc.FaceStream(oc, width_to_process=W, scale_factor_pct=10).process(pool[i])
            assert np.array_equal(res[i][0], eb) and np.array_equal(res[i][1], eid), (rep, i)
        with pytest.raises(capi.NvcaError):
            ctx.host_unregister(pool[0])                         # not registered any more
        # the same pages as pageable memory: views into the block
        small = pool.reshape(-1)[: 97 * 83].reshape(83, 97)
        g = synth.make_gray(97, 83, 77 + rep, "natural", [(10, 8, 60)])
        small[:] = g
        got = ctx.detect_raw(casc, small, 1.1, capi.HAAR_SCALE_IMAGE, (3, 3))
        exp = orc.detect_raw(oc, g, 1.1, orc.HAAR_SCALE_IMAGE, (3, 3))
        assert np.array_equal(np.asarray(got).reshape(-1, 4), np.asarray(exp).reshape(-1, 4)), rep
        big = pool.reshape(-1)[: 1100 * 2048].reshape(1100, 2048)
        big[:] = np.random.default_rng(rep).integers(0, 256, size=big.shape, dtype=np.uint8)
        assert np.array_equal(ctx.equalize_hist(big), orc.equalize_hist(big)), rep
        for i in range(N):                                       # put the frames back for the next turn
            pool[i] = synth.make_bgr(W, H, 1900 + i, "natural", [(20 + 10 * i, 30, 120)])
