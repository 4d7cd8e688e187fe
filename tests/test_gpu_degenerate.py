"""Degenerate content: constant frames (window variance 0: every threshold collapses to 0), 1-pixel checkerboards and
stripes (maximal variance), saturated noise.  Raw candidate lists and grouped boxes must still equal the oracle's -- also when more
raw candidates come out than the context's lists hold at first (the detectMultiScale entry points re-run with lists of the
exact size; only the batched face path may report NVCA_ERR_OVERFLOW for such a batch, and never truncates silently)."""
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
    exp_raw = orc.detect_raw(ocasc, img, 1.2, 0, (0, 0), cap=1 << 18)
    got_raw = ctx.detect_raw(casc, img, 1.2, 0, (0, 0), cap=1 << 18)           # lists beyond the context's capacity are re-run with room
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


def test_find_biggest_with_a_lenient_cascade_answers(env):
    """CV_HAAR_FIND_BIGGEST_OBJECT with a cascade that lets most windows through and minSize (1, 1) -- the nose / mouth / ear
    elements' call (NOSE/kmsnosedetect.cpp:870-873): OpenCV's sequential search stops at its first object; the product evaluates
    every ladder step first, which overflows the candidate lists -- it must re-run with room and return the reference's box."""
    import orc
    from nubovca import capi, synth
    ctx = env[0]
    xml = synth.synthetic_cascade_xml(seed=5, stages=[2, 3])
    # stage thresholds far below any vote sum: every window passes
    import re
    xml = re.sub(r"<stage_threshold>[^<]*</stage_threshold>", "<stage_threshold>-1000</stage_threshold>", xml)
    casc, oc = ctx.load_cascade_xml(xml), orc.parse_cascade_xml(xml)
    g = orc.equalize_hist(synth.make_gray(200, 150, 3, "natural", [(40, 20, 90)]))
    n_raw = len(orc.detect_raw(oc, g, 1.1, 0, (1, 1), cap=1 << 20))
    assert n_raw > 16384                                     # the first launch set cannot fit the default lists
    for flags in (capi.HAAR_FIND_BIGGEST_OBJECT, capi.HAAR_FIND_BIGGEST_OBJECT | capi.HAAR_DO_ROUGH_SEARCH):
        for mn in (3, 1):
            got = ctx.detect_multiscale(casc, g, 1.1, mn, flags, (1, 1))
            exp = orc.detect_multiscale(oc, g, 1.1, mn, flags, (1, 1))
            assert len(exp) == 1 and np.array_equal(got, exp), (flags, mn, got, exp)
    # and the plain scan of the same image: the complete raw list, in order
    assert np.array_equal(ctx.detect_raw(casc, g, 1.1, 0, (1, 1), cap=1 << 20), orc.detect_raw(oc, g, 1.1, 0, (1, 1), cap=1 << 20))
    # a nose stream on a frame whose face region is searched with this cascade: the element emits a nose, not an error
    face = ctx.load_cascade_xml(synth.synthetic_cascade_xml())
    oface = orc.parse_cascade_xml(synth.synthetic_cascade_xml())
    gs = capi.PartStream(ctx, 1, face, casc)
    os_ = orc.PartStream(1, oface, oc)
    seen = 0
    for i in range(3):
        f = synth.make_bgr(640, 480, 300 + i, "natural", [(150 + 6 * i, 90, 240)])
        ga, gb = gs.process(f)
        ea, eb = os_.process(f)
        assert np.array_equal(ga, ea) and np.array_equal(gb, eb), (i, ga, ea)
        seen += len(ea)
    assert seen > 0
