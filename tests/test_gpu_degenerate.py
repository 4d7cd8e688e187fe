"""Degenerate content: constant frames (window variance 0: every threshold collapses to 0), 1-pixel checkerboards and
stripes (maximal variance), saturated noise.  Raw candidate lists and grouped boxes must still equal the oracle's -- or the
call must fail loudly with NVCA_ERR_OVERFLOW when more raw candidates come out than the context's capacity holds (OpenCV
would group hundreds of thousands of rectangles there; the product never truncates silently)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(synth_xml):
    import orc
    from nubovca import capi
    ctx = capi.Context(0)
    yield ctx, ctx.load_cascade_xml(synth_xml), orc.parse_cascade_xml(synth_xml)
    ctx.close()


def _images(W, H):
    yy, xx = np.mgrid[0:H, 0:W]
    rng = np.random.default_rng(5)
    return {
        "black": np.zeros((H, W), np.uint8), "white": np.full((H, W), 255, np.uint8), "gray": np.full((H, W), 128, np.uint8),
        "checker1": (((xx + yy) & 1) * 255).astype(np.uint8), "checker8": ((((xx >> 3) + (yy >> 3)) & 1) * 255).astype(np.uint8),
        "vstripes": ((xx & 1) * 255).astype(np.uint8), "hstripes": ((yy & 1) * 255).astype(np.uint8),
        "saturated_noise": (rng.integers(0, 2, size=(H, W)) * 255).astype(np.uint8),
        "ramp": ((xx * 255) // max(W - 1, 1)).astype(np.uint8),
    }


@pytest.mark.parametrize("name", ["black", "white", "gray", "checker1", "checker8", "vstripes", "hstripes", "saturated_noise", "ramp"])
def test_degenerate_content_matches_oracle(env, name):
    import orc
    from nubovca import capi
    ctx, casc, ocasc = env
    img = _images(320, 240)[name]
    exp_raw = orc.detect_raw(ocasc, img, 1.2, 0, (0, 0))
    try:
        got_raw = ctx.detect_raw(casc, img, 1.2, 0, (0, 0), cap=1 << 18)
    except capi.NvcaError as e:
        assert e.code == capi.ERR_OVERFLOW and len(exp_raw) > 16384, (name, e.code, len(exp_raw))
        return
    assert np.array_equal(got_raw, exp_raw), (name, len(got_raw), len(exp_raw))
    assert np.array_equal(ctx.detect_multiscale(casc, img, 1.2, 3, 0, (24, 24)), orc.detect_multiscale(ocasc, img, 1.2, 3, 0, (24, 24))), name
    for flags in (capi.HAAR_SCALE_IMAGE, capi.HAAR_FIND_BIGGEST_OBJECT):
        assert np.array_equal(ctx.detect_multiscale(casc, img, 1.2, 2, flags, (20, 20)), orc.detect_multiscale(ocasc, img, 1.2, 2, flags, (20, 20))), (name, flags)


def test_degenerate_frames_through_the_face_batch(env):
    """the batched face path (equalizeHist of a constant image is the constant itself) on a mix of degenerate frames"""
    import orc
    from nubovca import capi
    ctx, casc, ocasc = env
    W, H = 640, 480
    imgs = _images(W, H)
    names = ["black", "checker8", "ramp", "gray", "saturated_noise", "white"]
    frames = [np.ascontiguousarray(np.repeat(imgs[n][:, :, None], 3, axis=2)) for n in names]
    streams = [capi.FaceStream(ctx, casc, width_to_process=W // 2, multi_scale_factor=20) for _ in names]
    oracles = [orc.FaceStream(ocasc, width_to_process=W // 2, scale_factor_pct=20) for _ in names]
    for t in range(2):
        try:
            res = ctx.face_batch_process(streams, [capi.make_frame(f) for f in frames])
        except capi.NvcaError as e:
            assert e.code == capi.ERR_OVERFLOW
            return
        for i, n in enumerate(names):
            eb, eid = oracles[i].process(frames[i])
            assert np.array_equal(res[i][0], eb) and np.array_equal(res[i][1], eid), (t, n)
